#!/usr/bin/env bash
# Probes the GPU box for any OpenCV (VERDICT r4 Next #2).  Output: gpurun_out/r05_opencv_probe.txt
out=gpurun_out/r05_opencv_probe.txt
mkdir -p gpurun_out
{
  echo "== python cv2"
  python3 -c "import cv2; print(cv2.__version__); print(cv2.getBuildInformation())" 2>&1 | head -40
  echo "== libopencv / opencv2 headers (find / -maxdepth 6)"
  find / -maxdepth 6 \( -name 'libopencv_core*' -o -name 'opencv2' -o -name 'opencv4' -o -name 'OpenCVConfig*.cmake' \) 2>/dev/null | head -40
  echo "== pkg-config"
  (pkg-config --modversion opencv4 || pkg-config --modversion opencv) 2>&1 | head -5
  echo "== pip list | grep -i opencv"
  python3 -m pip list 2>/dev/null | grep -i -E 'opencv|cv2|scikit-image|kornia|pillow|imageio' 
  echo "== other CV libs importable"
  for m in skimage kornia PIL imageio torchvision; do python3 -c "import $m; print('$m', getattr($m,'__version__','?'))" 2>&1 | tail -1; done
  echo "== host"
  nproc; grep -m1 'model name' /proc/cpuinfo; free -g | head -2
  echo "== gpus"
  python3 -c "import torch; print('device_count', torch.cuda.device_count())"
  rocm-smi --showtopo 2>&1 | head -30
} > "$out" 2>&1
cat "$out"

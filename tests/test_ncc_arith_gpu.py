"""The short exact sqrt / division of disparityNCorr (csrc/ncc_arith.hpp) against the compiler's correctly rounded
sqrtf and division, on the device: every float with exponent in [-64, 96] for the square root, 10^11 pairs
(random and adversarial mantissas) for the quotient.  tools/probes/ncc_arith_probe.hip includes the very header the
kernel uses; the rows it marks [used] must report no mismatch."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_ncc_arith_probe(tmp_path):
    exe = tmp_path / "ncc_probe"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-gpu-flush-denormals-to-zero",
                    os.path.join(ROOT, "tools", "probes", "ncc_arith_probe.hip"), "-o", str(exe)], check=True,
                   capture_output=True, timeout=600)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=600).stdout
    used = [l for l in out.splitlines() if "[used]" in l]
    assert len(used) == 4, out  # one sqrt row, three division modes
    for l in used:
        if l.startswith("sqrt"):
            m = re.search(r"exponent in \[-64, 96\] (\d+)", l)
        else:
            m = re.search(r"mismatches (\d+) of", l)
        assert m and int(m.group(1)) == 0, l
    # the probe is not vacuous: a sequence known to be wrong in places is reported as such
    assert any("rsq(p) as the reciprocal" in l and not re.search(r"mismatches 0 of", l) for l in out.splitlines()), out

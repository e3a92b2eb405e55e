// ps5_demo -- the ps5 driver's dense-flow step (denseLKWrapper, ProblemSets/ps5_cpp/src/Solution.cpp:40-84,
// as runProblem4 calls it :248-290) end to end on the shim, without OpenCV:
//   ps5_demo <prev.ppm|pgm|bmp> <next...> <out_dir> [window = 15] [naive|pyr = pyr]
//   ps5_demo --sequence <out_dir> <window> <frame0> <frame1> [<frame2> ...]   (all consecutive pairs, one library call)
// reads the two frames (colour or grey), runs lk::calcOpticalFlow / lk::calcOpticalFlowPyr through
// libmicv.so, writes <out_dir>/flow.ppm (arrows), flow-uColorMap.ppm, flow-vColorMap.ppm, u.f32, v.f32
// and pyramid.pgm (savePyramid of the previous frame's 4-level pyramid).
// Build: g++ -std=c++17 -O2 examples/ps5_demo.cpp -o ps5_demo -Lintrotocomputervision_amd -lmicv
//        -Wl,-rpath,$PWD/introtocomputervision_amd      (or link micv::shim from the top-level CMakeLists.txt)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../introtocomputervision_amd/shim/micv_viz.hpp"

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s prev.ppm next.ppm out_dir [window] [naive|pyr]\n", argv[0]);
        return 2;
    }
    try {
        if (std::string(argv[1]) == "--sequence") {
            // ps5_demo --sequence <out_dir> <window> <frame0> <frame1> <frame2> ...: the flows of all consecutive pairs
            // (runProblem4's loop, Solution.cpp:255-285) through the frame-sequence entry; u<p>.f32 / v<p>.f32 per pair
            if (argc < 6) {
                std::fprintf(stderr, "usage: %s --sequence out_dir window frame0 frame1 [frame2 ...]\n", argv[0]);
                return 2;
            }
            const std::string out = argv[2];
            const size_t win = (size_t)std::atoi(argv[3]);
            std::vector<micv_viz::Mat> frames;
            for (int i = 4; i < argc; i++) frames.push_back(micv_viz::imread(argv[i]));
            const auto flows = micv_viz::denseLKSequence(frames, win, out, "flow");
            for (size_t p = 0; p < flows.size(); p++)
                for (int k = 0; k < 2; k++) {
                    const micv_viz::Mat &m = k ? flows[p].second : flows[p].first;
                    std::ofstream f(out + (k ? "/v" : "/u") + std::to_string(p) + ".f32", std::ios::binary);
                    for (int y = 0; y < m.rows; y++) f.write(reinterpret_cast<const char *>(m.ptr<float>(y)), (std::streamsize)m.cols * 4);
                }
            std::printf("ps5_demo: %zu frames, %zu pairs, window %zu, 4-level pyramid -> %s/flow<p>.ppm\n", frames.size(), flows.size(), win, out.c_str());
            return 0;
        }
        const micv_viz::Mat prev = micv_viz::imread(argv[1]), next = micv_viz::imread(argv[2]);
        const std::string out = argv[3];
        const size_t win = argc > 4 ? (size_t)std::atoi(argv[4]) : 15;  // config/ps5.yaml lk_window_size_4
        const bool naive = argc > 5 && std::string(argv[5]) == "naive";
        micv_shim::log_kernel_times_to([](const std::string &l) { std::fprintf(stderr, "[info] %s\n", l.c_str()); });
        auto uv = micv_viz::denseLKWrapper(prev, next, naive ? micv_viz::LKMode::NAIVE : micv_viz::LKMode::HEIRARCHICAL,
                                           win, out, "flow");
        for (int k = 0; k < 2; k++) {
            const micv_viz::Mat &m = k ? uv.second : uv.first;
            std::ofstream f(out + (k ? "/v.f32" : "/u.f32"), std::ios::binary);
            for (int y = 0; y < m.rows; y++) f.write(reinterpret_cast<const char *>(m.ptr<float>(y)), (std::streamsize)m.cols * 4);
        }
        if ((prev.rows >> 3) > 0 && (prev.cols >> 3) > 0)
            micv_viz::savePyramid(pyr::makeGaussianPyramid(prev, 4), out + "/pyramid.pgm");  // Solution.cpp:182-184
        std::printf("ps5_demo: %dx%d, window %zu, %s -> %s/flow.ppm\n", prev.cols, prev.rows, win, naive ? "single level" : "4-level pyramid",
                    out.c_str());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ps5_demo: %s\n", e.what());
        return 1;
    }
    return 0;
}

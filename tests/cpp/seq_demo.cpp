// seq_demo.cpp -- the frame-sequence host entry through the C ABI (and the shim's wrapper), from C++:
// micv_lk_flow_seq_host over N frames against N - 1 calls of micv_lk_flow_pyr_frames_host, byte for byte.
//   g++ -std=c++17 tests/cpp/seq_demo.cpp -Lintrotocomputervision_amd -lmicv -o seq_demo && ./seq_demo 6 120 200 3 0
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../introtocomputervision_amd/shim/micv_shim.hpp"

static uint64_t sm64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 5, rows = argc > 2 ? atoi(argv[2]) : 96, cols = argc > 3 ? atoi(argv[3]) : 160;
    const int cn = argc > 4 ? atoi(argv[4]) : 1, f32 = argc > 5 ? atoi(argv[5]) : 1;
    const size_t es = f32 ? 4 : 1, fb = (size_t)rows * cols * cn * es;
    // a smooth pattern that drifts from frame to frame, plus noise
    std::vector<std::vector<unsigned char>> frames(n, std::vector<unsigned char>(fb));
    uint64_t seed = 12345;
    for (int t = 0; t < n; t++)
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++)
                for (int c = 0; c < cn; c++) {
                    const int v = (((x + 2 * t) / 7 + (y + t) / 5 + c) * 37 + (int)(sm64(seed) & 15)) & 255;
                    const size_t i = ((size_t)y * cols + x) * cn + c;
                    if (f32) reinterpret_cast<float *>(frames[t].data())[i] = (float)v + 0.25f * c;
                    else frames[t][i] = (unsigned char)v;
                }
    micv_ctx *ctx = nullptr;
    if (micv_ctx_create(0, &ctx) != MICV_OK) { fprintf(stderr, "ctx: %s\n", micv_last_error()); return 2; }
    const size_t ob = (size_t)rows * cols * 4;
    std::vector<std::vector<float>> su(n - 1, std::vector<float>((size_t)rows * cols)), sv = su, pu = su, pv = su;
    std::vector<const void *> fp;
    std::vector<float *> up, vp;
    for (int t = 0; t < n; t++) fp.push_back(frames[t].data());
    for (int p = 0; p + 1 < n; p++) { up.push_back(su[p].data()); vp.push_back(sv[p].data()); }
    const int depth = f32 ? MICV_DEPTH_32F : MICV_DEPTH_8U;
    if (micv_lk_flow_seq_host(ctx, fp.data(), n, rows, cols, (size_t)cols * cn * es, cn, depth, 15, 4, up.data(), vp.data(),
                              (size_t)cols * 4) != MICV_OK) { fprintf(stderr, "seq: %s\n", micv_last_error()); return 3; }
    int bad = 0;
    for (int p = 0; p + 1 < n; p++) {
        if (micv_lk_flow_pyr_frames_host(ctx, frames[p].data(), frames[p + 1].data(), rows, cols, (size_t)cols * cn * es, cn, depth,
                                         15, 4, pu[p].data(), pv[p].data(), (size_t)cols * 4) != MICV_OK) {
            fprintf(stderr, "pair: %s\n", micv_last_error());
            return 4;
        }
        if (memcmp(pu[p].data(), su[p].data(), ob) || memcmp(pv[p].data(), sv[p].data(), ob)) bad++;
    }
    // the shim's wrapper on micv_shim::Mat (its own context)
    std::vector<micv_shim::Mat> mf, mu, mv;
    for (int t = 0; t < n; t++) {
        mf.emplace_back(rows, cols, micv::make_type(f32 ? micv::CV_32F : micv::CV_8U, cn));
        memcpy(mf.back().data, frames[t].data(), fb);
    }
    lk::calcOpticalFlowPyrSequence(mf, mu, mv, 15, 4);
    for (int p = 0; p + 1 < n; p++)
        if (memcmp(mu[p].data, su[p].data(), ob) || memcmp(mv[p].data, sv[p].data(), ob)) bad++;
    // errors: one frame only, a null frame
    if (micv_lk_flow_seq_host(ctx, fp.data(), 1, rows, cols, (size_t)cols * cn * es, cn, depth, 15, 4, up.data(), vp.data(), (size_t)cols * 4) != MICV_EINVAL) bad += 100;
    fp[1] = nullptr;
    if (micv_lk_flow_seq_host(ctx, fp.data(), n, rows, cols, (size_t)cols * cn * es, cn, depth, 15, 4, up.data(), vp.data(), (size_t)cols * 4) != MICV_EINVAL) bad += 100;
    micv_ctx_destroy(ctx);
    printf("%s pairs=%d mismatching=%d\n", bad ? "FAIL" : "OK", n - 1, bad);
    return bad ? 1 : 0;
}

// micv_shim.hpp -- header-only drop-in for the reference's cv::Mat-in / cv::Mat-out functions,
// implemented on the C ABI of libmicv.so (include/mi_cv.h).
//
// Same namespaces, names, argument order, defaults and ownership rules as the reference:
//   lk::      ProblemSets/ps5_cpp/include/OpticalFlow.h:5-19
//   pyr::     ProblemSets/ps5_cpp/include/Pyramids.h:7-12
//   harris::  ProblemSets/ps4_cpp/include/Harris.h:18-96   (cpu:: and gpu:: are the same code here)
//   sift::    ProblemSets/ps4_cpp/include/Descriptors.h:8-23
//   cuda:: / serial::  ps2_cpp/include/DisparitySSD.h:18-43, DisparityNCorr.h:19-44,
//                      ps1_cpp/src/Hough.h:22-84
// Inputs are const references and never retained; outputs are (re)allocated by the callee like
// the reference does (Mat::create / assignment of a fresh Mat).  Errors: the reference asserts or
// exit(-1)s (CudaCommon.cuh:13-22); the shim throws std::runtime_error carrying micv_last_error().
//
// With -DMICV_SHIM_WITH_OPENCV the functions take real cv::Mat / cv::KeyPoint; without it they
// take the minimal micv::Mat of micv_mat.hpp (this image has no OpenCV).
#pragma once

#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mi_cv.h"

#ifdef MICV_SHIM_WITH_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/core/cuda.hpp>
#include <opencv2/core/types.hpp>
namespace micv_shim {
using Mat = cv::Mat;
using KeyPoint = cv::KeyPoint;
using Size = cv::Size;
// cv::cuda::GpuMat's data pointer must be memory this library's device can address (an OpenCV
// built against HIP, or unified memory); the overloads below only pass it through.
using GpuMat = cv::cuda::GpuMat;
enum { F32 = CV_32F, S8 = CV_8S, U8 = CV_8U, S32 = CV_32S };
inline micv_ctx *context() {  // one context per thread, like the reference's per-call streams
    thread_local micv_ctx *ctx = nullptr;
    if (!ctx && micv_ctx_create(0, &ctx) != MICV_OK)
        throw std::runtime_error(std::string("micv: ") + micv_last_error());
    return ctx;
}
inline void create_continuous(GpuMat &m, int rows, int cols, int type) { cv::cuda::createContinuous(rows, cols, type, m); }
}  // namespace micv_shim
#else
#include "micv_mat.hpp"
namespace micv_shim {
using Mat = micv::Mat;
using KeyPoint = micv::KeyPoint;
using Size = micv::Size;
using GpuMat = micv::GpuMat;
enum { F32 = micv::CV_32F, S8 = micv::CV_8S, U8 = micv::CV_8U, S32 = micv::CV_32S };
inline micv_ctx *context() { return micv::thread_context(); }
inline void create_continuous(GpuMat &m, int rows, int cols, int type) { m.create(rows, cols, type); }
}  // namespace micv_shim
#endif

namespace micv_shim {

// The reference's kernel-timing log lines (ps4_cpp/lib/Harris.cu:155,290, ps2_cpp/lib/DisparitySSD.cu:203,
// DisparityNCorr.cu:247, ps1_cpp/src/Hough.cu:289,345,391, ps5_cpp/lib/Pyramids.cu:69,123):
//     micv_shim::log_kernel_times_to([](const std::string &line) { spdlog::get("file_logger")->info(line); });
// makes every shim call of those functions emit "<kernel> execution took <ms> ms" ("<kernel> took <ms> ms"
// for the two pyramid kernels, as the reference words them) through the given sink; an empty function
// removes it.  One sink per process (the drivers log from their main thread only).
inline std::function<void(const std::string &)> &kernel_log_sink() {
    static std::function<void(const std::string &)> sink;
    return sink;
}
inline void log_kernel_times_to(std::function<void(const std::string &)> sink) {
    kernel_log_sink() = std::move(sink);
    if (!kernel_log_sink()) {
        micv_set_kernel_log(nullptr, nullptr);
        return;
    }
    micv_set_kernel_log(
        [](const char *kernel, float ms, void *) {
            const std::string k(kernel);
            const bool pyr = k.compare(0, 3, "pyr") == 0;
            if (kernel_log_sink()) kernel_log_sink()(k + (pyr ? " took " : " execution took ") + std::to_string(ms) + " ms");
        },
        nullptr);
}

inline void check(int rc) {
    if (rc != MICV_OK) throw std::runtime_error(std::string("micv: ") + micv_last_error());
}
inline void require(bool ok, const char *what) {
    if (!ok) throw std::invalid_argument(what);
}
// Multi-GPU (include/mi_cv.h, "multi-GPU"): one process per GPU.  With a communicator set, lk::calcOpticalFlowPyr on
// CV_32F frames splits every call by rows over the ranks (each computes its band, the coarse-flow halo travels over
// RCCL, the bands are gathered) and still returns whole fields -- the reference's caller (ps5_cpp/src/Solution.cpp:
// 60-64) does not change.  Rank 0 makes the id (micv_comm_unique_id) and ships its MICV_COMM_ID_BYTES bytes to the
// others by whatever the application has (MPI, a file); every rank then calls init_comm(id, rank, world) --
// collective -- or use_comm() with an ncclComm_t / micv_comm it already owns.
inline micv_comm *&comm_slot() {
    static micv_comm *c = nullptr;
    return c;
}
inline void use_comm(micv_comm *c) { comm_slot() = c; }
inline void init_comm(const void *unique_id, int rank, int world) {
    micv_comm *c = nullptr;
    check(micv_comm_create(context(), nullptr, unique_id, rank, world, &c));
    comm_slot() = c;
    check(micv_comm_selftest(context(), c, nullptr));  // a broken fabric is reported here, not as a wrong flow later
}
inline void close_comm() {
    if (comm_slot()) micv_comm_destroy(comm_slot());
    comm_slot() = nullptr;
}
inline bool frame_type_ok(const Mat &m) {  // what cvtColor(COLOR_RGB2GRAY) + convertTo(CV_32F) take here
    return (m.depth() == U8 || m.depth() == F32) && (m.channels() == 1 || m.channels() == 3 || m.channels() == 4);
}
inline int depth_code(const Mat &m) { return m.depth() == U8 ? MICV_DEPTH_8U : MICV_DEPTH_32F; }
// The reference converts its inputs itself: pyr::makeGaussianPyramid runs
// cv::cvtColor(COLOR_RGB2GRAY) on multi-channel frames and convertTo(CV_32F) (Pyramids.cpp:9-15),
// lk::calcOpticalFlow convertTo(CV_32F) (OpticalFlow.cpp:51-52).  Single-channel CV_32F passes
// through untouched; everything else goes through micv_to_gray_f32_host.
inline Mat to_f32(const Mat &m) {
    if (m.type() == F32) return m;
    require(frame_type_ok(m), "micv shim: 1/3/4-channel CV_8U or CV_32F input expected");
    Mat f(m.rows, m.cols, F32);
    check(micv_to_gray_f32_host(context(), m.data, m.rows, m.cols, m.step, m.channels(), depth_code(m),
                                f.ptr<float>(), f.step));
    return f;
}

}  // namespace micv_shim

namespace lk {
using micv_shim::Mat;

inline void calcOpticalFlow(const Mat &prevImg, const Mat &nextImg, Mat &u, Mat &v,
                            const size_t winSize = 21) {
    micv_shim::require(prevImg.rows == nextImg.rows && prevImg.cols == nextImg.cols &&
                           prevImg.type() == nextImg.type(),
                       "lk::calcOpticalFlow: size/type mismatch");  // OpticalFlow.cpp:47
    micv_shim::require(winSize % 2 == 1, "lk::calcOpticalFlow: winSize must be odd");  // :48
    const Mat p = micv_shim::to_f32(prevImg), n = micv_shim::to_f32(nextImg);
    micv_shim::require(p.step == n.step, "lk::calcOpticalFlow: inputs need equal row pitch");
    u.create(p.rows, p.cols, micv_shim::F32);  // :53-54
    v.create(p.rows, p.cols, micv_shim::F32);
    micv_shim::check(micv_lk_flow_host(micv_shim::context(), p.ptr<float>(), n.ptr<float>(), p.rows,
                                       p.cols, p.step, static_cast<int>(winSize), u.ptr<float>(),
                                       v.ptr<float>(), u.step));
}

inline void warp(const Mat &src, const Mat &du, const Mat &dv, Mat &dst) {
    micv_shim::require(du.rows == dv.rows && du.cols == dv.cols && src.rows == du.rows &&
                           src.cols == du.cols,
                       "lk::warp: size mismatch");  // OpticalFlow.cpp:107
    micv_shim::require(du.type() == micv_shim::F32 && dv.type() == micv_shim::F32 &&
                           src.type() == micv_shim::F32 && du.step == dv.step,
                       "lk::warp: CV_32F expected");  // :108
    Mat out(src.rows, src.cols, micv_shim::F32);
    micv_shim::check(micv_lk_warp_host(micv_shim::context(), src.ptr<float>(), src.step,
                                       du.ptr<float>(), dv.ptr<float>(), du.step, src.rows, src.cols,
                                       out.ptr<float>(), out.step));
    dst = out;
}

// An extension, not a reference function: the pyramid depth the reference hard-codes (pyrDepth = 4,
// OpticalFlow.cpp:127) as a parameter.  lk::calcOpticalFlowPyr below has the reference's exact type
// (tests/cpp/shim_signatures.cpp) and calls this with 4.
inline void calcOpticalFlowPyrLevels(const Mat &prevImg, const Mat &nextImg, Mat &u, Mat &v,
                                     const size_t winSize, const size_t levels) {
    micv_shim::require(prevImg.rows == nextImg.rows && prevImg.cols == nextImg.cols,
                       "lk::calcOpticalFlowPyr: size mismatch");
    Mat uu(prevImg.rows, prevImg.cols, micv_shim::F32), vv(prevImg.rows, prevImg.cols, micv_shim::F32);
    if (prevImg.type() == nextImg.type() && prevImg.type() != micv_shim::F32 && prevImg.step == nextImg.step) {
        // the ps5 driver's call (Solution.cpp:63): colour and/or 8-bit frames, converted by
        // makeGaussianPyramid (Pyramids.cpp:9-15) -- one upload, conversion on the device
        micv_shim::require(micv_shim::frame_type_ok(prevImg),
                           "lk::calcOpticalFlowPyr: 1/3/4-channel CV_8U or CV_32F frames expected");
        micv_shim::check(micv_lk_flow_pyr_frames_host(
            micv_shim::context(), prevImg.data, nextImg.data, prevImg.rows, prevImg.cols, prevImg.step,
            prevImg.channels(), micv_shim::depth_code(prevImg), static_cast<int>(winSize),
            static_cast<int>(levels), uu.ptr<float>(), vv.ptr<float>(), uu.step));
        u = uu;
        v = vv;
        return;
    }
    const Mat p = micv_shim::to_f32(prevImg), n = micv_shim::to_f32(nextImg);
    micv_shim::require(p.step == n.step, "lk::calcOpticalFlowPyr: inputs need equal row pitch");
    if (micv_shim::comm_slot())  // row-sharded over the ranks of the communicator, whole fields on every rank
        micv_shim::check(micv_lk_flow_pyr_rowshard_host(micv_shim::context(), micv_shim::comm_slot(), p.ptr<float>(),
                                                        n.ptr<float>(), p.rows, p.cols, p.step, static_cast<int>(winSize),
                                                        static_cast<int>(levels), uu.ptr<float>(), vv.ptr<float>(), uu.step));
    else
        micv_shim::check(micv_lk_flow_pyr_host(micv_shim::context(), p.ptr<float>(), n.ptr<float>(),
                                               p.rows, p.cols, p.step, static_cast<int>(winSize),
                                               static_cast<int>(levels), uu.ptr<float>(),
                                               vv.ptr<float>(), uu.step));
    u = uu;  // :165-166
    v = vv;
}
// ps5_cpp/include/OpticalFlow.h:14-18
inline void calcOpticalFlowPyr(const Mat &prevImg, const Mat &nextImg, Mat &u, Mat &v, const size_t winSize = 21) {
    calcOpticalFlowPyrLevels(prevImg, nextImg, u, v, winSize, 4);
}
// An extension for the driver's frame loops (ps5_cpp/src/Solution.cpp:255-285: pairs (0, 1), (1, 2), ... of the frames
// Config.cpp:17-46 loaded from a directory): the same flows as calling lk::calcOpticalFlowPyr pair by pair, with every
// frame uploaded once and transfers overlapped with the chains (micv_lk_flow_seq_host).  u[p], v[p]: pair (p, p + 1).
inline void calcOpticalFlowPyrSequence(const std::vector<Mat> &frames, std::vector<Mat> &u, std::vector<Mat> &v,
                                       const size_t winSize = 21, const size_t levels = 4) {
    micv_shim::require(frames.size() >= 2, "lk::calcOpticalFlowPyrSequence: at least two frames expected");
    const Mat &f0 = frames[0];
    micv_shim::require(micv_shim::frame_type_ok(f0), "lk::calcOpticalFlowPyrSequence: 1/3/4-channel CV_8U or CV_32F frames expected");
    std::vector<const void *> fp;
    for (const Mat &f : frames) {
        micv_shim::require(f.rows == f0.rows && f.cols == f0.cols && f.type() == f0.type() && f.step == f0.step,
                           "lk::calcOpticalFlowPyrSequence: frames of one size, type and row pitch expected");
        fp.push_back(f.data);
    }
    const size_t n = frames.size() - 1;
    std::vector<Mat> uu, vv;
    std::vector<float *> up, vp;
    for (size_t p = 0; p < n; p++) {
        uu.emplace_back(f0.rows, f0.cols, micv_shim::F32);
        vv.emplace_back(f0.rows, f0.cols, micv_shim::F32);
        up.push_back(uu.back().ptr<float>());
        vp.push_back(vv.back().ptr<float>());
    }
    micv_shim::check(micv_lk_flow_seq_host(micv_shim::context(), fp.data(), static_cast<int>(frames.size()), f0.rows, f0.cols,
                                           f0.step, f0.channels(), micv_shim::depth_code(f0), static_cast<int>(winSize),
                                           static_cast<int>(levels), up.data(), vp.data(), uu[0].step));
    u = uu;
    v = vv;
}
}  // namespace lk

namespace pyr {
using micv_shim::Mat;

inline void pyrDown(const Mat &src, Mat &dst) {
    micv_shim::require(src.type() == micv_shim::F32, "pyr::pyrDown: CV_32F expected");  // Pyramids.cu:35
    Mat out(src.rows / 2, src.cols / 2, micv_shim::F32);
    micv_shim::check(micv_pyr_down_host(micv_shim::context(), src.ptr<float>(), src.rows, src.cols,
                                        src.step, out.ptr<float>(), out.step));
    dst = out;
}
inline void pyrUp(const Mat &src, Mat &dst) {
    micv_shim::require(src.type() == micv_shim::F32, "pyr::pyrUp: CV_32F expected");  // Pyramids.cu:95
    Mat out(src.rows * 2, src.cols * 2, micv_shim::F32);  // may alias src: written after the read
    micv_shim::check(micv_pyr_up_host(micv_shim::context(), src.ptr<float>(), src.rows, src.cols,
                                      src.step, out.ptr<float>(), out.step));
    dst = out;
}
inline std::vector<Mat> makeGaussianPyramid(const Mat &src, const size_t levels) {
    const Mat grey = micv_shim::to_f32(src);  // Pyramids.cpp:9-15: cvtColor(RGB2GRAY) if colour, convertTo(CV_32F)
    std::vector<Mat> pyramid;
    std::vector<float *> ptrs;
    for (size_t l = 0; l < levels; l++) {
        pyramid.emplace_back(grey.rows >> l, grey.cols >> l, micv_shim::F32);
        ptrs.push_back(pyramid.back().ptr<float>());
    }
    micv_shim::check(micv_gaussian_pyramid_host(micv_shim::context(), grey.ptr<float>(), grey.rows,
                                                grey.cols, grey.step, static_cast<int>(levels),
                                                ptrs.data()));
    return pyramid;
}
}  // namespace pyr

namespace harris {
using micv_shim::Mat;

inline void getGradients(const Mat &in, int kernelSize, Mat &diffX, Mat &diffY) {
    micv_shim::require(kernelSize == 1 || kernelSize == 3 || kernelSize == 5 || kernelSize == 7,
                       "harris::getGradients: kernelSize must be 1, 3, 5 or 7");  // Harris.cpp:16
    micv_shim::require(in.type() == micv_shim::F32, "harris::getGradients: CV_32F expected");  // :17
    diffX.create(in.rows, in.cols, micv_shim::F32);
    diffY.create(in.rows, in.cols, micv_shim::F32);
    micv_shim::check(micv_sobel_host(micv_shim::context(), in.ptr<float>(), in.rows, in.cols, in.step,
                                     kernelSize, 1.f, diffX.ptr<float>(), diffY.ptr<float>(),
                                     diffX.step));
}

namespace detail {
inline void response(const Mat &gradX, const Mat &gradY, const size_t windowSize,
                     const double gaussianSigma, const float harrisScore, Mat &cornerResponse, int flags) {
    micv_shim::require(gradX.rows == gradY.rows && gradX.cols == gradY.cols &&
                           gradX.type() == micv_shim::F32 && gradY.type() == micv_shim::F32 &&
                           gradX.step == gradY.step,
                       "harris::getCornerResponse: gradient mismatch");  // Harris.cpp:49-50
    micv_shim::require(windowSize % 2 == 1, "harris::getCornerResponse: windowSize must be odd");
    Mat out(gradX.rows, gradX.cols, micv_shim::F32);
    micv_shim::check(micv_harris_response_ex_host(micv_shim::context(), gradX.ptr<float>(),
                                                  gradY.ptr<float>(), gradX.rows, gradX.cols, gradX.step,
                                                  static_cast<int>(windowSize), gaussianSigma,
                                                  harrisScore, flags, out.ptr<float>(), out.step));
    cornerResponse = out;
}
inline void refine(const Mat &cornerResponse, const double threshold, const int minDistance,
                   Mat &corners, std::vector<std::pair<int, int>> &cornerLocs, bool append) {
    micv_shim::require(cornerResponse.type() == micv_shim::F32,
                       "harris::refineCorners: CV_32F expected");  // Harris.cpp:105
    Mat out(cornerResponse.rows, cornerResponse.cols, micv_shim::F32);
    const int64_t cap = static_cast<int64_t>(cornerResponse.rows) * cornerResponse.cols;
    std::vector<int32_t> locs(static_cast<size_t>(cap) * 2);
    int64_t n = 0;
    micv_shim::check(micv_harris_refine_host(micv_shim::context(), cornerResponse.ptr<float>(),
                                             cornerResponse.rows, cornerResponse.cols,
                                             cornerResponse.step, threshold, minDistance,
                                             out.ptr<float>(), out.step, locs.data(), cap, &n));
    corners = out;
    if (!append) cornerLocs.clear();  // gpu:: resizes (Harris.cu:324), cpu:: appends (Harris.cpp:138)
    for (int64_t i = 0; i < n; i++) cornerLocs.emplace_back(locs[2 * i], locs[2 * i + 1]);
}
}  // namespace detail

namespace cpu {
// harris::cpu's own arithmetic (Harris.cpp:78-92: unfused accumulation, double determinant, one rounding),
// computed on the GPU: MICV_HARRIS_CPU.  -DMICV_SHIM_HARRIS_CPU_AS_GPU=1 routes it to harris::gpu's instead.
inline void getCornerResponse(const Mat &gradX, const Mat &gradY, const size_t windowSize,
                              const double gaussianSigma, const float harrisScore,
                              Mat &cornerResponse) {
#if defined(MICV_SHIM_HARRIS_CPU_AS_GPU) && MICV_SHIM_HARRIS_CPU_AS_GPU
    detail::response(gradX, gradY, windowSize, gaussianSigma, harrisScore, cornerResponse, 0);
#else
    detail::response(gradX, gradY, windowSize, gaussianSigma, harrisScore, cornerResponse, MICV_HARRIS_CPU);
#endif
}
inline void refineCorners(const Mat &cornerResponse, const double threshold, const int minDistance,
                          Mat &corners, std::vector<std::pair<int, int>> &cornerLocs) {
    detail::refine(cornerResponse, threshold, minDistance, corners, cornerLocs, true);
}
}  // namespace cpu
namespace gpu {
inline void getCornerResponse(const Mat &gradX, const Mat &gradY, const size_t windowSize,
                              const double gaussianSigma, const float harrisScore,
                              Mat &cornerResponse) {
    detail::response(gradX, gradY, windowSize, gaussianSigma, harrisScore, cornerResponse, 0);
}
inline void refineCorners(const Mat &cornerResponse, const double threshold, const int minDistance,
                          Mat &corners, std::vector<std::pair<int, int>> &cornerLocs) {
    detail::refine(cornerResponse, threshold, minDistance, corners, cornerLocs, false);
}
}  // namespace gpu
}  // namespace harris

namespace sift {
using micv_shim::KeyPoint;
using micv_shim::Mat;

inline void getAnglesFromGradients(const Mat &gradX, const Mat &gradY, Mat &angles) {
    micv_shim::require(gradX.rows == gradY.rows && gradX.cols == gradY.cols &&
                           gradX.type() == micv_shim::F32 && gradY.type() == micv_shim::F32 &&
                           gradX.step == gradY.step,
                       "sift::getAnglesFromGradients: gradient mismatch");  // Descriptors.cpp:9-10
    angles.create(gradX.rows, gradX.cols, micv_shim::F32);
    micv_shim::check(micv_sift_angles_host(micv_shim::context(), gradX.ptr<float>(),
                                           gradY.ptr<float>(), gradX.rows, gradX.cols, gradX.step,
                                           angles.ptr<float>(), angles.step));
}
inline void getKeypoints(const Mat &gradX, const Mat &gradY,
                         const std::vector<std::pair<int, int>> &cornerLocs, const size_t size,
                         std::vector<KeyPoint> &keypoints) {
    keypoints.clear();  // Descriptors.cpp:36
    std::vector<int32_t> locs;
    for (const auto &c : cornerLocs) {
        locs.push_back(c.first);
        locs.push_back(c.second);
    }
    std::vector<float> kp(cornerLocs.size() * 4);
    micv_shim::check(micv_sift_keypoints_host(micv_shim::context(), gradX.ptr<float>(),
                                              gradY.ptr<float>(), gradX.rows, gradX.cols, gradX.step,
                                              locs.data(), static_cast<int64_t>(cornerLocs.size()),
                                              static_cast<float>(size), kp.data()));
    for (size_t i = 0; i < cornerLocs.size(); i++)
        keypoints.emplace_back(kp[4 * i], kp[4 * i + 1], kp[4 * i + 2], kp[4 * i + 3], 0.f);  // :45
}
// The descriptor step of Solution::siftHelper (ps4_cpp/src/Solution.cpp:166-169): where the
// reference calls cv::xfeatures2d::SIFT::compute(img, keypoints, descriptors), a build without
// opencv_contrib calls this on the gradient fields it already has.  descriptors: CV_32F, one row of
// 128 values per keypoint (the layout of OpenCV's SIFT output, ready for the BFMatcher step).
inline void computeDescriptors(const Mat &gradX, const Mat &gradY, const std::vector<KeyPoint> &keypoints,
                               Mat &descriptors) {
    micv_shim::require(gradX.rows == gradY.rows && gradX.cols == gradY.cols &&
                           gradX.type() == micv_shim::F32 && gradY.type() == micv_shim::F32 &&
                           gradX.step == gradY.step,
                       "sift::computeDescriptors: gradient mismatch");
    std::vector<float> kp(keypoints.size() * 4);
    for (size_t i = 0; i < keypoints.size(); i++) {
        kp[4 * i] = keypoints[i].pt.x;
        kp[4 * i + 1] = keypoints[i].pt.y;
        kp[4 * i + 2] = keypoints[i].size;
        kp[4 * i + 3] = keypoints[i].angle;
    }
    Mat out(static_cast<int>(keypoints.size()), 128, micv_shim::F32);
    if (!keypoints.empty())
        micv_shim::check(micv_sift_descriptors_host(micv_shim::context(), gradX.ptr<float>(), gradY.ptr<float>(),
                                                    gradX.rows, gradX.cols, gradX.step, kp.data(),
                                                    static_cast<int64_t>(keypoints.size()), out.ptr<float>(),
                                                    out.step));
    descriptors = out;
}
}  // namespace sift

namespace micv_shim {
inline void disparity(bool ncc, const Mat &left, const Mat &right, const size_t windowRad,
                      const int minDisparity, const int maxDisparity, int flags, Mat &disparity) {
    require(left.type() == F32 && right.type() == F32 && left.rows == right.rows &&
                left.cols == right.cols && left.step == right.step,
            "disparity: CV_32FC1 pair of equal size expected");  // DisparitySSD.cpp:15
    disparity.create(left.rows, left.cols, S8);                   // :32
    auto fn = ncc ? micv_disparity_ncorr_host : micv_disparity_ssd_host;
    check(fn(context(), left.ptr<float>(), right.ptr<float>(), left.rows, left.cols, left.step,
             static_cast<int>(windowRad), minDisparity, maxDisparity, flags,
             disparity.ptr<int8_t>(), disparity.step));
}
}  // namespace micv_shim

// `cuda::` = the CUDA kernels as written (2r-column window, 5e6 cut-off); `serial::` = the CPU
// functions as written.  micv_disparity_*(flags = 0) is the corrected (2r+1)^2 definition.
// The CUDA kernels also keep ROLLING column sums down 40-row strips (DisparitySSD.cu:97-138), which
// round differently from fresh sums on non-integer images.  ps2's inputs are 8-bit images converted to
// float (ps2_cpp/src/main.cpp:21-48), where both give the same disparities, so the shim stays on the fast
// kernel; build with -DMICV_SHIM_STEREO_ROLLING=1 to get the strip-serial sums for arbitrary f32 input.
#ifndef MICV_SHIM_STEREO_ROLLING
#define MICV_SHIM_STEREO_ROLLING 0
#endif
namespace cuda {
using micv_shim::Mat;
constexpr int kStereoRolling = MICV_SHIM_STEREO_ROLLING ? MICV_STEREO_ROLLING : 0;
inline void disparitySSD(const Mat &left, const Mat &right, const size_t windowRad,
                         const int minDisparity, const int maxDisparity, Mat &disparity) {
    micv_shim::disparity(false, left, right, windowRad, minDisparity, maxDisparity,
                         MICV_STEREO_COLS_2R | MICV_STEREO_MIN_SSD_5E6 | kStereoRolling, disparity);
}
inline void disparityNCorr(const Mat &left, const Mat &right, const size_t windowRad,
                           const int minDisparity, const int maxDisparity, Mat &disparity) {
    micv_shim::disparity(true, left, right, windowRad, minDisparity, maxDisparity,
                         MICV_STEREO_COLS_2R | kStereoRolling, disparity);
}
inline void houghLinesAccumulate(const Mat &edgeMask, const unsigned int rhoBinSize,
                                 const unsigned int thetaBinSize, Mat &accumulator) {
    micv_shim::require(edgeMask.type() == micv_shim::U8, "cuda::houghLinesAccumulate: CV_8UC1 expected");
    int rb = 0, tb = 0;
    micv_shim::check(micv_hough_lines_dims(edgeMask.rows, edgeMask.cols, rhoBinSize, thetaBinSize, &rb, &tb));
    Mat acc(rb, tb, micv_shim::S32);
    micv_shim::check(micv_hough_lines_host(micv_shim::context(), edgeMask.ptr<uint8_t>(), edgeMask.rows,
                                           edgeMask.cols, edgeMask.step, rhoBinSize, thetaBinSize,
                                           acc.ptr<int32_t>()));
    accumulator = acc;
}
inline void houghCirclesAccumulate(const Mat &edgeMask, const size_t radius, Mat &accumulator) {
    micv_shim::require(edgeMask.type() == micv_shim::U8, "cuda::houghCirclesAccumulate: CV_8UC1 expected");
    Mat acc(edgeMask.rows, edgeMask.cols, micv_shim::S32);
    micv_shim::check(micv_hough_circles_host(micv_shim::context(), edgeMask.ptr<uint8_t>(),
                                             edgeMask.rows, edgeMask.cols, edgeMask.step,
                                             static_cast<unsigned>(radius), acc.ptr<int32_t>()));
    accumulator = acc;
}
inline void findLocalMaxima(const Mat &accumulator, const unsigned int numPeaks, const int threshold,
                            std::vector<std::pair<unsigned int, unsigned int>> &localMaxima) {
    micv_shim::require(accumulator.type() == micv_shim::S32 && accumulator.isContinuous(),
                       "cuda::findLocalMaxima: continuous CV_32SC1 expected");
    std::vector<uint32_t> peaks(static_cast<size_t>(numPeaks) * 2 + 2);
    int64_t n = 0;
    micv_shim::check(micv_hough_peaks_host(micv_shim::context(), accumulator.ptr<int32_t>(),
                                           accumulator.rows, accumulator.cols, numPeaks, threshold,
                                           peaks.data(), &n));
    for (int64_t i = 0; i < n; i++) localMaxima.emplace_back(peaks[2 * i], peaks[2 * i + 1]);  // appended, Hough.cu:413
}

// The cv::cuda::GpuMat overloads (ps1_cpp/src/Hough.h:22-25, 48-51, 73-75; Hough.cu:251-286,
// 311-340, 366-426): device-resident in and out, nothing crosses PCIe except the peak list.
using micv_shim::GpuMat;
inline void houghLinesAccumulate(const GpuMat &edgeMask, const unsigned int rhoBinSize,
                                 const unsigned int thetaBinSize, GpuMat &accumulator) {
    micv_shim::require(edgeMask.type() == micv_shim::U8, "cuda::houghLinesAccumulate: CV_8UC1 expected");  // Hough.cu:255
    int rb = 0, tb = 0;
    micv_shim::check(micv_hough_lines_dims(edgeMask.rows, edgeMask.cols, rhoBinSize, thetaBinSize, &rb, &tb));
    micv_shim::create_continuous(accumulator, rb, tb, micv_shim::S32);  // :263
    micv_shim::check(micv_hough_lines_dev(micv_shim::context(), edgeMask.ptr<uint8_t>(), edgeMask.rows,
                                          edgeMask.cols, edgeMask.step, rhoBinSize, thetaBinSize,
                                          accumulator.ptr<int32_t>(), nullptr));
}
inline void houghCirclesAccumulate(const GpuMat &edgeMask, const size_t radius, GpuMat &accumulator) {
    micv_shim::require(edgeMask.type() == micv_shim::U8, "cuda::houghCirclesAccumulate: CV_8UC1 expected");
    micv_shim::create_continuous(accumulator, edgeMask.rows, edgeMask.cols, micv_shim::S32);  // Hough.cu:318
    micv_shim::check(micv_hough_circles_dev(micv_shim::context(), edgeMask.ptr<uint8_t>(), edgeMask.rows,
                                            edgeMask.cols, edgeMask.step, static_cast<unsigned>(radius),
                                            accumulator.ptr<int32_t>(), nullptr));
}
inline void findLocalMaxima(const GpuMat &accumulator, const unsigned int numPeaks, const int threshold,
                            std::vector<std::pair<unsigned int, unsigned int>> &localMaxima) {
    micv_shim::require(accumulator.type() == micv_shim::S32 && accumulator.isContinuous(),
                       "cuda::findLocalMaxima: continuous CV_32SC1 expected");
    // peaks (2 x u32 each) followed by the int64 count, in one device block
    const size_t peak_bytes = (static_cast<size_t>(numPeaks) * 2 + 2) * sizeof(uint32_t);
    void *blk = nullptr;
    micv_shim::check(micv_device_malloc(micv_shim::context(), peak_bytes + 8, &blk));
    std::vector<uint32_t> host(peak_bytes / 4 + 2);
    int rc = micv_hough_peaks_dev(micv_shim::context(), accumulator.ptr<int32_t>(), accumulator.rows,
                                  accumulator.cols, numPeaks, threshold, static_cast<uint32_t *>(blk),
                                  reinterpret_cast<int64_t *>(static_cast<char *>(blk) + peak_bytes), nullptr);
    if (rc == MICV_OK)
        rc = micv_memcpy2d_d2h(micv_shim::context(), host.data(), peak_bytes + 8, blk, peak_bytes + 8,
                               peak_bytes + 8, 1);
    micv_device_free(micv_shim::context(), blk);
    micv_shim::check(rc);
    int64_t n = 0;
    std::memcpy(&n, reinterpret_cast<const char *>(host.data()) + peak_bytes, 8);
    for (int64_t i = 0; i < n; i++) localMaxima.emplace_back(host[2 * i], host[2 * i + 1]);  // appended, Hough.cu:413
}
}  // namespace cuda

namespace serial {
using micv_shim::Mat;
inline void disparitySSD(const Mat &left, const Mat &right, const size_t windowRad,
                         const int minDisparity, const int maxDisparity, Mat &disparity) {
    micv_shim::disparity(false, left, right, windowRad, minDisparity, maxDisparity,
                         MICV_STEREO_SERIAL, disparity);
}
// serial::disparityNCorr calls cv::matchTemplate per pixel (DisparityNCorr.cpp:60-64), whose
// arithmetic is OpenCV's; the shim routes it to the (2r+1)^2 NCC defined in DESIGN.md.
inline void disparityNCorr(const Mat &left, const Mat &right, const size_t windowRad,
                           const int minDisparity, const int maxDisparity, Mat &disparity) {
    micv_shim::disparity(true, left, right, windowRad, minDisparity, maxDisparity, 0, disparity);
}
}  // namespace serial

// ---- "next" rows -----------------------------------------------------------------

namespace mhi {  // ProblemSets/ps7_cpp/include/MotionHistory.h:7-28 (single-channel CV_8U frames)
using micv_shim::Mat;
using micv_shim::Size;
inline void frameDifference(const Mat &f1, const Mat &f2, const double thresh, Mat &diff,
                            const Size &blurSize = Size(3, 3), const double blurSigma = 1.0) {  // MotionHistory.h:10-15
    micv_shim::require(f1.type() == micv_shim::U8 && f2.type() == micv_shim::U8 && f1.rows == f2.rows &&
                           f1.cols == f2.cols && f1.step == f2.step,
                       "mhi::frameDifference: two CV_8UC1 frames of one size expected");
    Mat out(f1.rows, f1.cols, micv_shim::U8);
    micv_shim::check(micv_mhi_frame_difference_host(micv_shim::context(), f1.ptr<uint8_t>(), f2.ptr<uint8_t>(),
                                                    f1.rows, f1.cols, f1.step, thresh, blurSize.width,
                                                    blurSize.height, blurSigma, out.ptr<uint8_t>(), out.step));
    diff = out;
}
inline void calcMotionHistory(Mat &history, const Mat &binaryMask, const int tau) {
    micv_shim::require(history.type() == micv_shim::U8 && binaryMask.type() == micv_shim::U8 &&
                           history.rows == binaryMask.rows && history.cols == binaryMask.cols,
                       "mhi::calcMotionHistory: CV_8UC1 history and mask of one size expected");
    micv_shim::check(micv_mhi_update_host(micv_shim::context(), history.ptr<uint8_t>(), history.step,
                                          binaryMask.ptr<uint8_t>(), binaryMask.step, history.rows,
                                          history.cols, tau));
}
inline void energyFromHistory(const Mat &mhi, Mat &mei) {  // MotionHistory.h:23, MotionHistory.cpp:98-105
    micv_shim::require(mhi.type() == micv_shim::U8, "mhi::energyFromHistory: CV_8UC1 expected");
    Mat out(mhi.rows, mhi.cols, micv_shim::U8);
    micv_shim::check(micv_mhi_energy_host(micv_shim::context(), mhi.ptr<uint8_t>(), mhi.rows, mhi.cols, mhi.step,
                                          out.ptr<uint8_t>(), out.step));
    mei = out;
}
inline void energyFromHistory(const std::vector<Mat> &mhis, std::vector<Mat> &meis) {  // :27, .cpp:107-112
    for (const auto &m : mhis) {
        Mat mei;
        energyFromHistory(m, mei);
        meis.push_back(mei);  // appended, like the reference
    }
}
}  // namespace mhi

namespace sol {
using micv_shim::Mat;
// sol::generateEdge, ps1_cpp/src/Solution.cpp:21-47, with Config::EdgeDetect spelled out
// (gaussianSize, gaussianSigma, lowerThreshold, upperThreshold; Sobel aperture 3).
inline void generateEdge(const Mat &input, const int gaussianSize, const double gaussianSigma,
                         const double lowerThreshold, const double upperThreshold, Mat &output) {
    micv_shim::require(input.type() == micv_shim::U8, "sol::generateEdge: CV_8UC1 expected");
    Mat out(input.rows, input.cols, micv_shim::U8);
    micv_shim::check(micv_generate_edge_host(micv_shim::context(), input.ptr<uint8_t>(), input.rows, input.cols,
                                             input.step, gaussianSize, gaussianSigma, lowerThreshold,
                                             upperThreshold, out.ptr<uint8_t>(), out.step));
    output = out;
}
// The matching step of Solution::siftHelper, ps4_cpp/src/Solution.cpp:172-184:
// BFMatcher::knnMatch(d1, d2, raw, 2) + `m[0].distance < ratio * m[1].distance`.
// goodMatches receives (queryIdx, trainIdx) pairs, distances the matching m[0].distance.
inline void matchDescriptors(const Mat &d1, const Mat &d2, const double ratio,
                             std::vector<std::pair<int, int>> &goodMatches, std::vector<float> &distances) {
    micv_shim::require(d1.type() == micv_shim::F32 && d2.type() == micv_shim::F32 && d1.cols == d2.cols &&
                           d2.rows >= 2,
                       "sol::matchDescriptors: CV_32FC1 descriptor matrices of one width expected");
    std::vector<int32_t> idx2(static_cast<size_t>(d1.rows) * 2), m(static_cast<size_t>(d1.rows) * 2);
    std::vector<float> dist2(static_cast<size_t>(d1.rows) * 2), dd(static_cast<size_t>(d1.rows));
    micv_shim::check(micv_bf_knn2_host(micv_shim::context(), d1.ptr<float>(), d1.rows, d1.step, d2.ptr<float>(),
                                       d2.rows, d2.step, d1.cols, idx2.data(), dist2.data()));
    int64_t n = 0;
    micv_shim::check(micv_bf_ratio_filter_host(micv_shim::context(), idx2.data(), dist2.data(), d1.rows, ratio,
                                               m.data(), dd.data(), d1.rows, &n));
    goodMatches.clear();
    distances.clear();
    for (int64_t i = 0; i < n; i++) {
        goodMatches.emplace_back(m[2 * i], m[2 * i + 1]);
        distances.push_back(dd[i]);
    }
}
}  // namespace sol

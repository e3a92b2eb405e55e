// micv_mat.hpp -- the handful of cv::Mat / cv::cuda::GpuMat / cv::KeyPoint members the shim
// touches, for builds WITHOUT OpenCV (this image has none).  Same member names and meaning as
// OpenCV's, so micv_shim.hpp compiles unchanged against either; define MICV_SHIM_WITH_OPENCV to use
// the real classes.  This is NOT a stand-in for building the reference: it exists so the shim's
// plumbing (sizes, steps, channels, (re)allocation, ownership) can be compiled and tested here.
#pragma once
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mi_cv.h"

namespace micv {

// OpenCV's type codes: depth in the low 3 bits, (channels - 1) above them (CV_MAKETYPE).
enum { CV_8U = 0, CV_8S = 1, CV_32S = 4, CV_32F = 5 };
constexpr int make_type(int depth, int cn) { return (depth & 7) + ((cn - 1) << 3); }
enum {
    CV_8UC1 = CV_8U, CV_8UC3 = make_type(CV_8U, 3), CV_8UC4 = make_type(CV_8U, 4),
    CV_32FC1 = CV_32F, CV_32FC3 = make_type(CV_32F, 3), CV_32FC4 = make_type(CV_32F, 4)
};
inline int type_depth(int type) { return type & 7; }
inline int type_channels(int type) { return (type >> 3) + 1; }
inline int elem_size(int type) {  // bytes per pixel (all channels)
    const int d = type_depth(type);
    return ((d == CV_8U || d == CV_8S) ? 1 : 4) * type_channels(type);
}

struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size &o) const { return width == o.width && height == o.height; }
};

// One context per host thread, shared by the shim's functions and by GpuMat's allocator.
inline micv_ctx *thread_context() {
    thread_local micv_ctx *ctx = nullptr;
    if (!ctx && micv_ctx_create(0, &ctx) != MICV_OK)
        throw std::runtime_error(std::string("micv: ") + micv_last_error());
    return ctx;
}

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;  // bytes per row
    unsigned char *data = nullptr;

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    // non-owning view over caller memory (cv::Mat(rows, cols, type, data, step))
    Mat(int r, int c, int type, void *ptr, size_t step_bytes = 0)
        : rows(r), cols(c), step(step_bytes ? step_bytes : (size_t)c * elem_size(type)),
          data(static_cast<unsigned char *>(ptr)), type_(type) {}

    void create(int r, int c, int type) {
        if (data && r == rows && c == cols && type == type_ && owner_) return;
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * elem_size(type);
        owner_ = std::shared_ptr<unsigned char>(new unsigned char[step * (size_t)r + 16],
                                                std::default_delete<unsigned char[]>());
        data = owner_.get();
    }
    void create(Size s, int type) { create(s.height, s.width, type); }
    static Mat zeros(int r, int c, int type) {
        Mat m(r, c, type);
        std::memset(m.data, 0, m.step * (size_t)r);
        return m;
    }
    int type() const { return type_; }
    int depth() const { return type_depth(type_); }
    int channels() const { return type_channels(type_); }
    size_t elemSize() const { return (size_t)elem_size(type_); }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elem_size(type_); }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int y = 0; y < rows; y++) std::memcpy(m.data + y * m.step, data + y * step, m.step);
        return m;
    }
    template <typename T>
    T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (size_t)y * step); }
    template <typename T>
    const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (size_t)y * step); }
    template <typename T>
    T &at(int y, int x) { return ptr<T>(y)[x]; }
    template <typename T>
    const T &at(int y, int x) const { return ptr<T>(y)[x]; }

private:
    int type_ = CV_8U;
    std::shared_ptr<unsigned char> owner_;
};

// cv::cuda::GpuMat's members as the reference uses them (Hough.cu:251-364: create, upload,
// download, rows / cols / step / data / type).  Device memory comes from the C ABI
// (micv_device_malloc / micv_memcpy2d_*), dense rows (step = cols * elemSize).
class GpuMat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char *data = nullptr;

    GpuMat() = default;
    GpuMat(int r, int c, int type) { create(r, c, type); }
    explicit GpuMat(const Mat &m) { upload(m); }
    void create(int r, int c, int type) {
        if (data && r == rows && c == cols && type == type_) return;
        micv_ctx *ctx = thread_context();
        void *p = nullptr;
        if (micv_device_malloc(ctx, (size_t)r * c * elem_size(type), &p) != MICV_OK)
            throw std::runtime_error(std::string("micv: ") + micv_last_error());
        owner_ = std::shared_ptr<unsigned char>(static_cast<unsigned char *>(p),
                                                [ctx](unsigned char *q) { micv_device_free(ctx, q); });
        data = owner_.get();
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * elem_size(type);
    }
    void create(Size s, int type) { create(s.height, s.width, type); }
    void upload(const Mat &m) {
        create(m.rows, m.cols, m.type());
        if (micv_memcpy2d_h2d(thread_context(), data, step, m.data, m.step, step, rows) != MICV_OK)
            throw std::runtime_error(std::string("micv: ") + micv_last_error());
    }
    void download(Mat &m) const {
        m.create(rows, cols, type_);
        if (micv_memcpy2d_d2h(thread_context(), m.data, m.step, data, step, step, rows) != MICV_OK)
            throw std::runtime_error(std::string("micv: ") + micv_last_error());
    }
    int type() const { return type_; }
    int depth() const { return type_depth(type_); }
    int channels() const { return type_channels(type_); }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elem_size(type_); }
    template <typename T>
    T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (size_t)y * step); }
    template <typename T>
    const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (size_t)y * step); }

private:
    int type_ = CV_8U;
    std::shared_ptr<unsigned char> owner_;
};

struct Point2f {
    float x = 0, y = 0;
};
struct KeyPoint {  // cv::KeyPoint(x, y, size, angle, response)
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
    KeyPoint() = default;
    KeyPoint(float x, float y, float s, float a = -1, float r = 0) : size(s), angle(a), response(r) {
        pt.x = x;
        pt.y = y;
    }
};

}  // namespace micv

// micv_mat.hpp -- the handful of cv::Mat / cv::KeyPoint members the shim touches, for builds
// WITHOUT OpenCV (this image has none).  Same member names and meaning as OpenCV's, so
// micv_shim.hpp compiles unchanged against either; define MICV_SHIM_WITH_OPENCV to use the
// real cv::Mat.  This is NOT a stand-in for building the reference: it exists so the shim's
// plumbing (sizes, steps, (re)allocation, ownership) can be compiled and tested here.
#pragma once
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace micv {

enum { CV_8U = 0, CV_8S = 1, CV_32S = 4, CV_32F = 5 };
inline int elem_size(int type) { return (type == CV_8U || type == CV_8S) ? 1 : 4; }

struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size &o) const { return width == o.width && height == o.height; }
};

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
    int channels() const { return 1; }
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

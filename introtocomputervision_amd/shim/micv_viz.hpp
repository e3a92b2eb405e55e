// micv_viz.hpp -- the driver-level plumbing of ps5 without OpenCV:
// image files in and out, drawVelocityVectors, the min-max normalisation + JET colour maps of
// denseLKWrapper, savePyramid.  Header-only, host code, no kernels (a frame has 900 arrows); it exists so a
// `ps5`-style demo runs end to end on the shim (examples/ps5_demo.cpp).  Reference:
//   drawVelocityVectors   ProblemSets/ps5_cpp/src/Solution.cpp:13-37
//   denseLKWrapper        ProblemSets/ps5_cpp/src/Solution.cpp:40-84
//   savePyramid           ProblemSets/ps5_cpp/src/Solution.cpp:86-101
// The reference delegates the pixel work to OpenCV 3.4.1 (cv::imread / imwrite, cv::arrowedLine,
// cv::normalize, cv::applyColorMap, cv::resize INTER_NEAREST, hconcat / vconcat), which is not in this
// image: what is restated here is OpenCV's published behaviour, PARITY UNPINNED like the rest of the
// library-call semantics:
//   * files: binary PGM (P5) / PPM (P6) and uncompressed 8 / 24 / 32-bit BMP instead of PNG (no zlib
//     here); a colour image is held B, G, R like cv::imread delivers it;
//   * cv::line, thickness 1, LINE_8: the integer Bresenham walk of cv::LineIterator (left to right,
//     err = dx - 2 dy, count = max(dx, dy) + 1), each pixel bounds-checked (OpenCV clips the segment to
//     the image first; the drawn pixels inside the image are the same walk);
//   * cv::arrowedLine(pt1, pt2, tipLength 0.1): the shaft, then two tip strokes from
//     pt2 + tip * (cos, sin)(angle +- pi/4), angle = atan2(pt1.y - pt2.y, pt1.x - pt2.x), cvRound-ed;
//   * cv::normalize(NORM_MINMAX, 0, 255, CV_8U): scale = 255 / (max - min) (0 when max - min <= DBL_EPSILON),
//     shift = -min * scale, dst = saturate_cast<uchar>(cvRound(src * (float)scale + (float)shift));
//   * cv::applyColorMap(COLORMAP_JET): the jet ramp r/g/b = clamp(1.5 - |4 x - 3 / 2 / 1|, 0, 1) at
//     x = i / 255 (OpenCV interpolates a 64-entry table of the same ramp; the two agree to within 2 of 255).
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "micv_shim.hpp"

namespace micv_viz {

using micv_shim::Mat;

struct Scalar {
    double v[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {}
};
struct Point {
    int x = 0, y = 0;
};

inline int cv_round(double v) { return (int)std::lrint(v); }  // finite, small arguments only
inline unsigned char sat_u8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ---- image files ------------------------------------------------------------------------------------
inline void skip_pnm_space(std::istream &f) {
    for (;;) {
        int c = f.peek();
        if (c == '#') {
            std::string line;
            std::getline(f, line);
        } else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') {
            f.get();
        } else {
            return;
        }
    }
}

// cv::imread(path, IMREAD_UNCHANGED) for P5 / P6 (maxval <= 255) and uncompressed BMP: CV_8UC1 or CV_8UC3 (B, G, R).
inline Mat imread(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("imread: cannot open " + path);
    char m0 = 0, m1 = 0;
    f.get(m0);
    f.get(m1);
    if (m0 == 'P' && (m1 == '5' || m1 == '6')) {
        int w = 0, h = 0, maxv = 0;
        skip_pnm_space(f);
        f >> w;
        skip_pnm_space(f);
        f >> h;
        skip_pnm_space(f);
        f >> maxv;
        f.get();  // the single whitespace byte before the raster
        if (!f || w <= 0 || h <= 0 || maxv <= 0 || maxv > 255) throw std::runtime_error("imread: bad PNM header in " + path);
        const int cn = m1 == '6' ? 3 : 1;
        Mat img(h, w, cn == 3 ? micv::CV_8UC3 : micv::CV_8UC1);
        std::vector<unsigned char> row((size_t)w * cn);
        for (int y = 0; y < h; y++) {
            f.read(reinterpret_cast<char *>(row.data()), (std::streamsize)row.size());
            if (!f) throw std::runtime_error("imread: truncated raster in " + path);
            unsigned char *d = img.ptr<unsigned char>(y);
            if (cn == 1) {
                std::copy(row.begin(), row.end(), d);
            } else {
                for (int x = 0; x < w; x++) {  // file order R, G, B -> B, G, R
                    d[3 * x] = row[3 * x + 2];
                    d[3 * x + 1] = row[3 * x + 1];
                    d[3 * x + 2] = row[3 * x];
                }
            }
        }
        return img;
    }
    if (m0 == 'B' && m1 == 'M') {
        unsigned char hdr[52];
        f.read(reinterpret_cast<char *>(hdr), 52);
        if (!f) throw std::runtime_error("imread: truncated BMP header in " + path);
        auto u32 = [&](int o) { return (unsigned)hdr[o] | ((unsigned)hdr[o + 1] << 8) | ((unsigned)hdr[o + 2] << 16) | ((unsigned)hdr[o + 3] << 24); };
        auto u16 = [&](int o) { return (unsigned)hdr[o] | ((unsigned)hdr[o + 1] << 8); };
        const unsigned off = u32(8), dib = u32(12);
        const int w = (int)u32(16), hs = (int)u32(20);
        const unsigned bpp = u16(26), comp = u32(28);
        unsigned ncol = u32(44);
        if (dib < 40 || w <= 0 || hs == 0 || comp != 0 || (bpp != 8 && bpp != 24 && bpp != 32))
            throw std::runtime_error("imread: unsupported BMP (need uncompressed 8 / 24 / 32 bit) " + path);
        const int h = hs < 0 ? -hs : hs;
        std::vector<unsigned char> pal;
        bool grey = false;
        if (bpp == 8) {
            if (ncol == 0) ncol = 256;
            pal.resize(4 * ncol);
            f.seekg(14 + dib);
            f.read(reinterpret_cast<char *>(pal.data()), (std::streamsize)pal.size());
            grey = true;
            for (unsigned i = 0; i < ncol; i++) grey = grey && pal[4 * i] == pal[4 * i + 1] && pal[4 * i] == pal[4 * i + 2];
        }
        const int cn = (bpp == 8 && grey) ? 1 : 3;
        Mat img(h, w, cn == 3 ? micv::CV_8UC3 : micv::CV_8UC1);
        const size_t rb = (((size_t)w * bpp + 31) / 32) * 4;
        std::vector<unsigned char> row(rb);
        f.seekg(off);
        for (int r = 0; r < h; r++) {
            f.read(reinterpret_cast<char *>(row.data()), (std::streamsize)rb);
            if (!f) throw std::runtime_error("imread: truncated BMP raster in " + path);
            unsigned char *d = img.ptr<unsigned char>(hs < 0 ? r : h - 1 - r);  // bottom-up unless the height is negative
            for (int x = 0; x < w; x++) {
                if (bpp == 8) {
                    const unsigned char *p = &pal[4 * (size_t)std::min<unsigned>(row[x], ncol - 1)];
                    if (cn == 1) d[x] = p[0];
                    else { d[3 * x] = p[0]; d[3 * x + 1] = p[1]; d[3 * x + 2] = p[2]; }
                } else {
                    const unsigned char *p = &row[(size_t)x * (bpp / 8)];  // stored B, G, R (, A)
                    d[3 * x] = p[0]; d[3 * x + 1] = p[1]; d[3 * x + 2] = p[2];
                }
            }
        }
        return img;
    }
    throw std::runtime_error("imread: " + path + " is neither P5 / P6 nor BMP");
}

// cv::imwrite for CV_8UC1 / CV_8UC3 (B, G, R): .pgm / .ppm by content, .bmp by extension.
inline void imwrite(const std::string &path, const Mat &img) {
    if (img.depth() != micv::CV_8U || (img.channels() != 1 && img.channels() != 3))
        throw std::invalid_argument("imwrite: CV_8UC1 or CV_8UC3 expected");
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("imwrite: cannot open " + path);
    const int w = img.cols, h = img.rows, cn = img.channels();
    const bool bmp = path.size() >= 4 && path.compare(path.size() - 4, 4, ".bmp") == 0;
    if (bmp) {
        const size_t rb = (((size_t)w * 24 + 31) / 32) * 4;
        unsigned char hdr[54] = {'B', 'M'};
        auto put32 = [&](int o, unsigned v) { hdr[o] = v & 255; hdr[o + 1] = (v >> 8) & 255; hdr[o + 2] = (v >> 16) & 255; hdr[o + 3] = (v >> 24) & 255; };
        put32(2, (unsigned)(54 + rb * h)); put32(10, 54); put32(14, 40); put32(18, (unsigned)w); put32(22, (unsigned)h);
        hdr[26] = 1; hdr[28] = 24; put32(34, (unsigned)(rb * h));
        f.write(reinterpret_cast<char *>(hdr), 54);
        std::vector<unsigned char> row(rb, 0);
        for (int y = h - 1; y >= 0; y--) {
            const unsigned char *s = img.ptr<unsigned char>(y);
            for (int x = 0; x < w; x++)
                for (int c = 0; c < 3; c++) row[3 * x + c] = cn == 3 ? s[3 * x + c] : s[x];
            f.write(reinterpret_cast<char *>(row.data()), (std::streamsize)rb);
        }
        return;
    }
    f << (cn == 3 ? "P6\n" : "P5\n") << w << " " << h << "\n255\n";
    std::vector<unsigned char> row((size_t)w * cn);
    for (int y = 0; y < h; y++) {
        const unsigned char *s = img.ptr<unsigned char>(y);
        if (cn == 1) std::copy(s, s + w, row.begin());
        else
            for (int x = 0; x < w; x++) { row[3 * x] = s[3 * x + 2]; row[3 * x + 1] = s[3 * x + 1]; row[3 * x + 2] = s[3 * x]; }
        f.write(reinterpret_cast<char *>(row.data()), (std::streamsize)row.size());
    }
}

// cv::cvtColor(COLOR_GRAY2RGB): replicate the channel.
inline Mat gray2rgb(const Mat &g) {
    Mat out(g.rows, g.cols, micv::CV_8UC3);
    for (int y = 0; y < g.rows; y++) {
        const unsigned char *s = g.ptr<unsigned char>(y);
        unsigned char *d = out.ptr<unsigned char>(y);
        for (int x = 0; x < g.cols; x++) d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = s[x];
    }
    return out;
}

// ---- drawing -----------------------------------------------------------------------------------------
inline void put_pixel(Mat &img, int x, int y, const Scalar &c) {
    if ((unsigned)x >= (unsigned)img.cols || (unsigned)y >= (unsigned)img.rows) return;
    unsigned char *d = img.ptr<unsigned char>(y) + (size_t)x * img.channels();
    for (int k = 0; k < img.channels(); k++) d[k] = sat_u8(cv_round(c.v[k]));
}

// cv::line(img, p1, p2, color): thickness 1, LINE_8 = cv::LineIterator's walk (left to right).
inline void line(Mat &img, Point p1, Point p2, const Scalar &color) {
    if (p1.x > p2.x) std::swap(p1, p2);
    int dx = p2.x - p1.x, dy = p2.y - p1.y;
    const int sy = dy < 0 ? -1 : 1;
    dy = dy < 0 ? -dy : dy;
    const bool steep = dy > dx;  // the major axis takes one step per pixel
    const int major = steep ? dy : dx, minor = steep ? dx : dy;
    int err = major - 2 * minor, x = p1.x, y = p1.y;
    for (int i = 0; i <= major; i++) {
        put_pixel(img, x, y, color);
        const bool both = err < 0;
        err += both ? 2 * major - 2 * minor : -2 * minor;
        if (steep) { y += sy; if (both) x += 1; }
        else { x += 1; if (both) y += sy; }
    }
}

// cv::arrowedLine(img, pt1, pt2, color) with the defaults thickness 1, LINE_8, shift 0, tipLength 0.1.
inline void arrowed_line(Mat &img, float x1, float y1, float x2, float y2, const Scalar &color) {
    const Point p1{cv_round(x1), cv_round(y1)}, p2{cv_round(x2), cv_round(y2)};  // Point2f -> Point: saturate_cast<int>
    const double ddx = (double)p1.x - p2.x, ddy = (double)p1.y - p2.y;
    const double tip = std::sqrt(ddx * ddx + ddy * ddy) * 0.1;
    line(img, p1, p2, color);
    const double angle = std::atan2(ddy, ddx), q = 3.14159265358979323846 / 4;
    Point p{cv_round(p2.x + tip * std::cos(angle + q)), cv_round(p2.y + tip * std::sin(angle + q))};
    line(img, p, p2, color);
    p = Point{cv_round(p2.x + tip * std::cos(angle - q)), cv_round(p2.y + tip * std::sin(angle - q))};
    line(img, p, p2, color);
}

// drawVelocityVectors (Solution.cpp:13-37): a 30 x 30 lattice of arrows (x, y) -> (x + u, y + v).
// Images under 30 pixels make the reference's strides 0 (its loop would not end); they are 1 here.
// Non-finite flow values are skipped (cv::Point2f -> Point of a NaN is undefined in OpenCV).
inline void drawVelocityVectors(Mat &inputImg, const Mat &u, const Mat &v, const Scalar &color) {
    micv_shim::require(u.rows == v.rows && u.cols == v.cols && u.rows == inputImg.rows && u.cols == inputImg.cols &&
                           u.type() == micv_shim::F32 && v.type() == micv_shim::F32 && inputImg.depth() == micv_shim::U8,
                       "drawVelocityVectors: image and CV_32FC1 flow fields of equal size expected");
    if (inputImg.channels() < 3) inputImg = gray2rgb(inputImg);
    constexpr int ARROWS_PER_RC = 30;
    const int rowStride = std::max(1, u.rows / ARROWS_PER_RC), colStride = std::max(1, u.cols / ARROWS_PER_RC);
    for (int y = 0; y < u.rows; y += rowStride)
        for (int x = 0; x < u.cols; x += colStride) {
            const float uVal = u.at<float>(y, x), vVal = v.at<float>(y, x);
            if (!std::isfinite(uVal) || !std::isfinite(vVal) || std::fabs(uVal) > 1e6f || std::fabs(vVal) > 1e6f) continue;
            arrowed_line(inputImg, (float)x, (float)y, (float)x + uVal, (float)y + vVal, color);
        }
}

// cv::normalize(src, dst, 0, 255, NORM_MINMAX, CV_8U) of a CV_32FC1 field (NaNs are ignored by the min / max).
inline Mat normalize_minmax_u8(const Mat &src) {
    micv_shim::require(src.type() == micv_shim::F32, "normalize: CV_32FC1 expected");
    double lo = DBL_MAX, hi = -DBL_MAX;
    for (int y = 0; y < src.rows; y++)
        for (int x = 0; x < src.cols; x++) {
            const float t = src.at<float>(y, x);
            if (t < lo) lo = t;
            if (t > hi) hi = t;
        }
    const double scale = 255.0 * (hi - lo > DBL_EPSILON ? 1.0 / (hi - lo) : 0.0), shift = 0.0 - lo * scale;
    const float a = (float)scale, b = (float)shift;
    Mat dst(src.rows, src.cols, micv::CV_8UC1);
    for (int y = 0; y < src.rows; y++)
        for (int x = 0; x < src.cols; x++) {
            const float t = src.at<float>(y, x) * a + b;
            dst.at<unsigned char>(y, x) = std::isfinite(t) ? sat_u8((int)std::lrintf(t)) : 0;
        }
    return dst;
}

// cv::applyColorMap(src, dst, COLORMAP_JET): CV_8UC1 -> CV_8UC3 (B, G, R).
inline Mat apply_colormap_jet(const Mat &src) {
    micv_shim::require(src.type() == micv::CV_8UC1, "applyColorMap: CV_8UC1 expected");
    unsigned char lut[256][3];
    for (int i = 0; i < 256; i++) {
        const double x = i / 255.0;
        auto ramp = [](double t) { return t < 0 ? 0.0 : (t > 1 ? 1.0 : t); };
        const double r = ramp(1.5 - std::fabs(4 * x - 3)), g = ramp(1.5 - std::fabs(4 * x - 2)), b = ramp(1.5 - std::fabs(4 * x - 1));
        lut[i][0] = sat_u8(cv_round(b * 255));
        lut[i][1] = sat_u8(cv_round(g * 255));
        lut[i][2] = sat_u8(cv_round(r * 255));
    }
    Mat dst(src.rows, src.cols, micv::CV_8UC3);
    for (int y = 0; y < src.rows; y++)
        for (int x = 0; x < src.cols; x++) {
            const unsigned char *c = lut[src.at<unsigned char>(y, x)];
            unsigned char *d = dst.ptr<unsigned char>(y) + 3 * x;
            d[0] = c[0]; d[1] = c[1]; d[2] = c[2];
        }
    return dst;
}

// ---- ps5 driver pieces ---------------------------------------------------------------------------------
enum class LKMode { NAIVE, HEIRARCHICAL };  // sic, Solution.cpp:11

// denseLKWrapper (Solution.cpp:40-84); image files get `ext` (".ppm") instead of ".png".
inline std::pair<Mat, Mat> denseLKWrapper(const Mat &prevImg, const Mat &nextImg, const LKMode mode, const size_t windowSize,
                                          const std::string &filePrefix, const std::string &outputImg,
                                          bool saveColorMaps = true, const std::string &ext = ".ppm") {
    Mat u, v;
    if (mode == LKMode::NAIVE) {
        // Solution.cpp:48-61: colour frames are reduced with cv::cvtColor(COLOR_RGB2GRAY) first; the shim's
        // lk::calcOpticalFlow takes CV_8U / CV_32F, 1 / 3 / 4 channels and converts the same way
        lk::calcOpticalFlow(prevImg, nextImg, u, v, windowSize);
    } else {
        lk::calcOpticalFlowPyr(prevImg, nextImg, u, v, windowSize);  // :63, the colour frames as they are
    }
    Mat velocityVectors = prevImg.clone();
    drawVelocityVectors(velocityVectors, u, v, Scalar(0, 255, 0, 255));
    imwrite(filePrefix + "/" + outputImg + ext, velocityVectors);
    if (saveColorMaps) {
        imwrite(filePrefix + "/" + outputImg + "-uColorMap" + ext, apply_colormap_jet(normalize_minmax_u8(u)));
        imwrite(filePrefix + "/" + outputImg + "-vColorMap" + ext, apply_colormap_jet(normalize_minmax_u8(v)));
    }
    return std::make_pair(u.clone(), v.clone());
}

// The driver's frame loops (runProblem4, Solution.cpp:255-285: denseLKWrapper on frames (0, 1), (1, 2), ... of a
// directory): the hierarchical flows of all consecutive pairs in ONE call of the frame-sequence entry (every frame
// uploaded once, transfers beside the chains), then denseLKWrapper's drawing per pair.  Files: <prefix>/<name><p><ext>.
inline std::vector<std::pair<Mat, Mat>> denseLKSequence(const std::vector<Mat> &frames, const size_t windowSize,
                                                        const std::string &filePrefix, const std::string &outputImg,
                                                        bool saveColorMaps = true, const std::string &ext = ".ppm") {
    std::vector<Mat> u, v;
    lk::calcOpticalFlowPyrSequence(frames, u, v, windowSize);
    std::vector<std::pair<Mat, Mat>> out;
    for (size_t p = 0; p < u.size(); p++) {
        const std::string name = outputImg + std::to_string(p);
        Mat velocityVectors = frames[p].clone();
        drawVelocityVectors(velocityVectors, u[p], v[p], Scalar(0, 255, 0, 255));
        imwrite(filePrefix + "/" + name + ext, velocityVectors);
        if (saveColorMaps) {
            imwrite(filePrefix + "/" + name + "-uColorMap" + ext, apply_colormap_jet(normalize_minmax_u8(u[p])));
            imwrite(filePrefix + "/" + name + "-vColorMap" + ext, apply_colormap_jet(normalize_minmax_u8(v[p])));
        }
        out.emplace_back(u[p], v[p]);
    }
    return out;
}

// cv::resize(src, dst, size, fx, fy, INTER_NEAREST) for an integer up-scale factor k.
inline Mat resize_nearest(const Mat &src, int rows, int cols) {
    Mat dst(rows, cols, src.type());
    const size_t es = src.elemSize();
    const double fy = (double)src.rows / rows, fx = (double)src.cols / cols;
    for (int y = 0; y < rows; y++) {
        const int sy = std::min((int)std::floor(y * fy), src.rows - 1);
        for (int x = 0; x < cols; x++) {
            const int sx = std::min((int)std::floor(x * fx), src.cols - 1);
            std::memcpy(dst.data + y * dst.step + x * es, src.data + sy * src.step + sx * es, es);
        }
    }
    return dst;
}

// savePyramid (Solution.cpp:86-101): the four levels scaled back to level-0 size, tiled 2 x 2; CV_32FC1
// levels are written after the min-max normalisation (cv::imwrite would saturate them).
inline void savePyramid(const std::vector<Mat> &pyramid, const std::string &filename) {
    micv_shim::require(pyramid.size() >= 4, "savePyramid: four levels expected");
    const int R = pyramid[0].rows, C = pyramid[0].cols;
    Mat all(2 * R, 2 * C, micv::CV_8UC1);
    for (int k = 0; k < 4; k++) {
        Mat lvl = pyramid[k].type() == micv_shim::F32 ? normalize_minmax_u8(pyramid[k]) : pyramid[k];
        Mat big = k == 0 ? lvl : resize_nearest(lvl, R, C);
        for (int y = 0; y < R; y++)
            std::memcpy(all.ptr<unsigned char>((k / 2) * R + y) + (k % 2) * C, big.ptr<unsigned char>(y), (size_t)C);
    }
    imwrite(filename, all);
}

}  // namespace micv_viz

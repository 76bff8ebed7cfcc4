// TEST INFRASTRUCTURE -- the slice of OpenCV that redtail's stereoDNN/sample_app/main.cpp uses (main.cpp:83-98, 317-330),
// so that the reference application compiles UNTOUCHED against libnvstereo_inference.so in an image without OpenCV:
// cv::Mat (8U / 16U / 32F, interleaved channels), imread (8-bit RGB / grey PNG via zlib), convertTo, resize(INTER_AREA,
// shrinking), cvtColor(BGR2RGB), reshape, t(), scalar *= and /=, imwrite (8 / 16-bit grey PNG).  Semantics follow
// OpenCV's documentation; the INTER_AREA filter restates computeResizeAreaTab (modules/imgproc/src/resize.cpp), like
// oracle/stereo_oracle.py:_area_table does -- "parity unpinned" in the same sense.  Never part of the product.
#pragma once
#include <zlib.h>

#include <algorithm>   // the real opencv.hpp pulls these in, and main.cpp relies on it (std::find_if, std::stringstream)
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_16U 2
#define CV_16S 3
#define CV_32F 5
#define CV_BGR2RGB 4

namespace cv {

typedef unsigned char uchar;
enum { INTER_AREA = 3 };
struct Size {
    int width, height;
    Size(int w = 0, int h = 0) : width(w), height(h) {}
};

class Mat {
public:
    int rows = 0, cols = 0;
    uchar* data = nullptr;

    Mat() {}
    Mat(int r, int c, int depth, int cn = 1) { create(r, c, depth, cn); }
    Mat(int r, int c, int depth, void* ext) : rows(r), cols(c), data(static_cast<uchar*>(ext)), depth_(depth), cn_(1) {}   // wraps, no copy
    explicit Mat(const std::vector<float>& v) : rows((int)v.size()), cols(1), data(reinterpret_cast<uchar*>(const_cast<float*>(v.data()))), depth_(CV_32F), cn_(1) {}

    void create(int r, int c, int depth, int cn) {
        rows = r; cols = c; depth_ = depth; cn_ = cn;
        store_ = std::make_shared<std::vector<uchar>>((size_t)r * c * cn * esz(depth));
        data = store_->data();
    }
    int depth() const { return depth_; }
    int channels() const { return cn_; }
    size_t total() const { return (size_t)rows * cols; }
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data) + (size_t)r * cols * cn_; }
    template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data) + (size_t)r * cols * cn_; }

    // saturate_cast<ushort>(cvRound(x)) for 32F -> 16U; plain widening for 8U -> 32F (what main.cpp needs)
    void convertTo(Mat& dst, int depth) const {
        Mat out(rows, cols, depth, cn_);
        const size_t n = total() * cn_;
        for (size_t i = 0; i < n; i++) {
            const double v = depth_ == CV_8U ? data[i] : depth_ == CV_16U ? reinterpret_cast<const uint16_t*>(data)[i]
                                                                          : reinterpret_cast<const float*>(data)[i];
            if (depth == CV_32F) reinterpret_cast<float*>(out.data)[i] = (float)v;
            else if (depth == CV_16U) reinterpret_cast<uint16_t*>(out.data)[i] = (uint16_t)std::min(65535.0, std::max(0.0, std::nearbyint(v)));
            else out.data[i] = (uchar)std::min(255.0, std::max(0.0, std::nearbyint(v)));
        }
        dst = out;
    }
    // same data, other channel count / row count (continuous matrices only)
    Mat reshape(int cn, int new_rows) const {
        Mat m = *this;
        const size_t elems = total() * cn_;
        m.cn_ = cn; m.rows = new_rows; m.cols = (int)(elems / cn / new_rows);
        return m;
    }
    Mat t() const {
        assert(depth_ == CV_32F && cn_ == 1);
        Mat out(cols, rows, CV_32F, 1);
        const float* s = ptr<float>();
        float* d = out.ptr<float>();
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) d[(size_t)c * rows + r] = s[(size_t)r * cols + c];
        return out;
    }
    Mat& operator*=(double k) { return scale(k); }
    Mat& operator/=(double k) { return scale(1.0 / k, k); }

private:
    static size_t esz(int depth) { return depth == CV_8U ? 1 : (depth == CV_16U || depth == CV_16S) ? 2 : 4; }
    Mat& scale(double k, double div = 0) {
        assert(depth_ == CV_32F);
        float* p = reinterpret_cast<float*>(data);
        const size_t n = total() * cn_;
        for (size_t i = 0; i < n; i++) p[i] = div != 0 ? (float)(p[i] / div) : (float)(p[i] * k);
        return *this;
    }
    int depth_ = CV_8U, cn_ = 1;
    std::shared_ptr<std::vector<uchar>> store_;
};

// ---- PNG (8-bit RGB / grey in, 8 / 16-bit grey out) ------------------------------------------------------------------------
namespace detail {
inline uint32_t be32(const uchar* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
inline void put32(std::vector<uchar>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uchar)(x >> s)); }
inline void chunk(std::vector<uchar>& out, const char* type, const std::vector<uchar>& body) {
    put32(out, (uint32_t)body.size());
    std::vector<uchar> tb(type, type + 4);
    tb.insert(tb.end(), body.begin(), body.end());
    out.insert(out.end(), tb.begin(), tb.end());
    put32(out, (uint32_t)crc32(0, tb.data(), (uInt)tb.size()));
}
}  // namespace detail

inline Mat imread(const std::string& filename) {
    FILE* f = fopen(filename.c_str(), "rb");
    if (!f) return Mat();
    std::vector<uchar> raw;
    uchar buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) raw.insert(raw.end(), buf, buf + n);
    fclose(f);
    static const uchar sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (raw.size() < 8 || memcmp(raw.data(), sig, 8) != 0) return Mat();
    int w = 0, h = 0, bits = 0, color = 0, interlace = 0;
    std::vector<uchar> idat;
    for (size_t off = 8; off + 12 <= raw.size();) {
        const uint32_t len = detail::be32(&raw[off]);
        const std::string type(reinterpret_cast<const char*>(&raw[off + 4]), 4);
        const uchar* body = &raw[off + 8];
        if (type == "IHDR") { w = (int)detail::be32(body); h = (int)detail::be32(body + 4); bits = body[8]; color = body[9]; interlace = body[12]; }
        else if (type == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (type == "IEND") break;
        off += 12 + len;
    }
    if (bits != 8 || (color != 2 && color != 0) || interlace != 0) return Mat();    // 8-bit RGB or grey, as the sample images
    const int cn = color == 2 ? 3 : 1, stride = w * cn;
    std::vector<uchar> px((size_t)h * (stride + 1));
    uLongf dlen = (uLongf)px.size();
    if (uncompress(px.data(), &dlen, idat.data(), (uLong)idat.size()) != Z_OK || dlen != px.size()) return Mat();
    Mat img(h, w, CV_8U, 3);
    std::vector<uchar> prev(stride, 0), cur(stride);
    for (int y = 0; y < h; y++) {
        const uchar* row = &px[(size_t)y * (stride + 1)];
        const int ft = row[0];
        for (int i = 0; i < stride; i++) {
            const int a = i >= cn ? cur[i - cn] : 0, b = prev[i], c = i >= cn ? prev[i - cn] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            cur[i] = (uchar)(row[1 + i] + pred);
        }
        uchar* d = img.ptr<uchar>(y);
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) d[3 * x + c] = cn == 3 ? cur[3 * x + 2 - c] : cur[x];      // RGB file -> BGR matrix
        prev = cur;
    }
    return img;
}

inline bool imwrite(const std::string& filename, const Mat& img) {
    if (img.channels() != 1 || (img.depth() != CV_8U && img.depth() != CV_16U)) return false;
    const int bps = img.depth() == CV_16U ? 2 : 1;
    std::vector<uchar> rawpx;
    for (int y = 0; y < img.rows; y++) {
        rawpx.push_back(0);
        for (int x = 0; x < img.cols; x++) {
            if (bps == 2) { const uint16_t v = img.ptr<uint16_t>(y)[x]; rawpx.push_back((uchar)(v >> 8)); rawpx.push_back((uchar)v); }
            else rawpx.push_back(img.ptr<uchar>(y)[x]);
        }
    }
    std::vector<uchar> z(compressBound((uLong)rawpx.size()));
    uLongf zl = (uLongf)z.size();
    if (compress2(z.data(), &zl, rawpx.data(), (uLong)rawpx.size(), 6) != Z_OK) return false;
    z.resize(zl);
    std::vector<uchar> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a}, ihdr;
    detail::put32(ihdr, (uint32_t)img.cols); detail::put32(ihdr, (uint32_t)img.rows);
    ihdr.push_back((uchar)(8 * bps)); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    detail::chunk(out, "IHDR", ihdr);
    detail::chunk(out, "IDAT", z);
    detail::chunk(out, "IEND", {});
    FILE* f = fopen(filename.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

// ---- geometry / colour -----------------------------------------------------------------------------------------------------
namespace detail {
struct Tap { int src; double w; };
// per destination index the (source index, weight) list of OpenCV's area filter for scale >= 1
inline std::vector<std::vector<Tap>> area_table(int ssize, int dsize) {
    const double scale = (double)ssize / dsize;
    std::vector<std::vector<Tap>> tab(dsize);
    for (int dx = 0; dx < dsize; dx++) {
        const double f1 = dx * scale, f2 = f1 + scale, cell = std::min(scale, ssize - f1);
        int s1 = (int)std::ceil(f1), s2 = std::min((int)std::floor(f2), ssize);
        s1 = std::min(s1, s2);
        if (s1 - f1 > 1e-3) tab[dx].push_back({s1 - 1, (s1 - f1) / cell});
        for (int sx = s1; sx < s2; sx++) tab[dx].push_back({sx, 1.0 / cell});
        if (f2 - s2 > 1e-3 && s2 < ssize) tab[dx].push_back({s2, std::min(std::min(f2 - s2, 1.0), cell) / cell});
    }
    return tab;
}
}  // namespace detail

inline void resize(const Mat& src, Mat& dst, Size sz, double, double, int interpolation) {
    assert(interpolation == INTER_AREA && src.depth() == CV_32F);
    assert(sz.width <= src.cols && sz.height <= src.rows && "INTER_AREA: shrinking only");
    (void)interpolation;
    const int cn = src.channels();
    if (sz.width == src.cols && sz.height == src.rows) { Mat c = src; dst = c; return; }
    const auto ty = detail::area_table(src.rows, sz.height), tx = detail::area_table(src.cols, sz.width);
    std::vector<double> rows((size_t)sz.height * src.cols * cn, 0.0);
    for (int y = 0; y < sz.height; y++)
        for (const auto& t : ty[y]) {
            const float* s = src.ptr<float>(t.src);
            double* d = &rows[(size_t)y * src.cols * cn];
            for (int i = 0; i < src.cols * cn; i++) d[i] += t.w * s[i];
        }
    Mat out(sz.height, sz.width, CV_32F, cn);
    for (int y = 0; y < sz.height; y++) {
        const double* s = &rows[(size_t)y * src.cols * cn];
        float* d = out.ptr<float>(y);
        for (int x = 0; x < sz.width; x++)
            for (int c = 0; c < cn; c++) {
                double acc = 0;
                for (const auto& t : tx[x]) acc += t.w * s[(size_t)t.src * cn + c];
                d[x * cn + c] = (float)acc;
            }
    }
    dst = out;
}

// fp32 <-> IEEE half stored as CV_16S (tests_main.cpp:200-201, 237-238); round to nearest even, no _Float16 in g++ 11
namespace detail {
inline uint16_t f2h(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                    // overflow -> inf
    if (x < 0x38800000u) {                                                      // subnormal half or zero
        if (x < 0x33000000u) return (uint16_t)sign;
        const int shift = 126 - (int)(x >> 23);
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const uint32_t r = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        m >>= shift;
        if (r > half || (r == half && (m & 1))) m++;
        return (uint16_t)(sign | m);
    }
    uint32_t m = x - 0x38000000u;
    const uint32_t r = m & 0x1fffu;
    m >>= 13;
    if (r > 0x1000u || (r == 0x1000u && (m & 1))) m++;
    return (uint16_t)(sign | m);
}
inline float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31, m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int s = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; s++; } x = sign | ((uint32_t)(113 - s) << 23) | ((mm & 0x3ffu) << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f; std::memcpy(&f, &x, 4);
    return f;
}
}  // namespace detail
inline void convertFp16(const Mat& src, Mat& dst) {
    const size_t n = src.total() * src.channels();
    if (src.depth() == CV_32F) {
        Mat out(src.rows, src.cols, CV_16S, src.channels());
        for (size_t i = 0; i < n; i++) out.ptr<uint16_t>()[i] = detail::f2h(src.ptr<float>()[i]);
        dst = out;
    } else {
        Mat out(src.rows, src.cols, CV_32F, src.channels());
        for (size_t i = 0; i < n; i++) out.ptr<float>()[i] = detail::h2f(src.ptr<uint16_t>()[i]);
        dst = out;
    }
}

inline void cvtColor(const Mat& src, Mat& dst, int code) {
    assert(code == CV_BGR2RGB && src.channels() == 3 && src.depth() == CV_32F);
    (void)code;
    Mat out(src.rows, src.cols, CV_32F, 3);
    for (int y = 0; y < src.rows; y++) {
        const float* s = src.ptr<float>(y);
        float* d = out.ptr<float>(y);
        for (int x = 0; x < src.cols; x++) { d[3 * x] = s[3 * x + 2]; d[3 * x + 1] = s[3 * x + 1]; d[3 * x + 2] = s[3 * x]; }
    }
    dst = out;
}

}  // namespace cv

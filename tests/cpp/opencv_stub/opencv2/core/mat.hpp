// NOT OpenCV.  A minimal stand-in for the handful of cv:: names include/dinov2_compat.hpp touches under
// DINOV2_WITH_OPENCV (cv::Mat rows/cols/data/type/create/clone/release/empty/isContinuous/size, cv::Size, CV_8UC3 / CV_32F /
// CV_32FC3), so that branch of the shim can be COMPILED AND RUN in an image that has no OpenCV.  Test infrastructure only
// (tests/test_gguf_and_abi.py::test_cpp_compat_opencv_branch_*); never shipped, never on the product's include path.
#pragma once
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)

namespace cv {

struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
};

class Mat {
public:
    // like cv::MatSize: callable (-> Size) and indexable ([0] = rows, [1] = cols), as inference.cpp:42,50 use it
    struct MatSize {
        const Mat* m;
        Size operator()() const { return Size(m->cols, m->rows); }
        int operator[](int i) const { return i == 0 ? m->rows : m->cols; }
    };
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;
    MatSize size{this};

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(const Mat& o) : rows(o.rows), cols(o.cols), data(o.data), size{this}, type_(o.type_), buf_(o.buf_) {}
    Mat& operator=(const Mat& o) {
        rows = o.rows; cols = o.cols; data = o.data; type_ = o.type_; buf_ = o.buf_;
        return *this;
    }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type;
        buf_ = std::make_shared<std::vector<unsigned char>>((size_t)r * c * elemSize());
        data = buf_->data();
    }
    void release() { rows = cols = 0; data = nullptr; buf_.reset(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { return (size_t)channels() * ((type_ & 7) == CV_32F ? 4 : 1); }
    bool isContinuous() const { return true; }
    Mat clone() const {
        Mat m(rows, cols, type_);
        if (data) std::memcpy(m.data, data, (size_t)rows * cols * elemSize());
        return m;
    }

private:
    int type_ = 0;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};

}  // namespace cv

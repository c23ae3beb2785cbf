// cvmat_stub.h — the three cv::Mat fields process_and_pad_image reads (rtpose.cpp:239-269).  TEST INFRASTRUCTURE.
#pragma once
namespace cv {
struct Mat {
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
};
}  // namespace cv

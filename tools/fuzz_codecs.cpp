// Mutation fuzzer for csrc/codecs.cpp (JPEG / PNG decoders) under ASan + UBSan.  From the repo root:
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 tools/fuzz_codecs.cpp -lz -o /tmp/fuzz_codecs && ASAN_OPTIONS=detect_leaks=0 /tmp/fuzz_codecs SEED ITERS
// Round 1: 320k iterations clean after bounds checks in the DQT/SOF/DRI/SOS parsers, Huffman-table validation and a 64 Mpixel cap.
#include "../caffe_rtpose_amd/csrc/codecs.cpp"
#include <random>
#include <dirent.h>
int main(int argc, char** argv) {
  std::vector<std::vector<unsigned char>> files;
  const char* dir = argc > 3 ? argv[3] : "tests/golden/codecs";
  DIR* d = opendir(dir);
  while (dirent* e = readdir(d)) {
    std::string n = e->d_name;
    if (n.size() > 4 && (n.substr(n.size() - 4) == ".jpg" || n.substr(n.size() - 4) == ".png")) {
      std::vector<unsigned char> b;
      read_file((std::string(dir) + "/" + n).c_str(), &b);
      files.push_back(b);
    }
  }
  std::mt19937 rng(argc > 1 ? atoi(argv[1]) : 1);
  const int N = argc > 2 ? atoi(argv[2]) : 20000;
  long ok = 0, bad = 0;
  std::vector<unsigned char> out;
  for (int it = 0; it < N; ++it) {
    std::vector<unsigned char> b = files[rng() % files.size()];
    const int mode = rng() % 100;
    if (mode < 50) { for (int k = 0, m = 1 + rng() % 8; k < m; ++k) b[rng() % b.size()] = (unsigned char)rng(); }
    else if (mode < 70) b.resize(1 + rng() % b.size());
    else if (mode < 85) { size_t i = rng() % b.size(); for (int k = 0, m = 1 + rng() % 16; k < m; ++k) b.insert(b.begin() + i, (unsigned char)rng()); }
    else { size_t i = rng() % (b.size() - 2); const unsigned char ms[] = {0xC0, 0xC2, 0xC4, 0xDA, 0xDB, 0xDD, 0xD9, 0xD0, 0xC9}; b[i] = 0xFF; b[i + 1] = ms[rng() % 9]; }
    int w = 0, h = 0;
    if (rtp_decode_image(b.data(), b.size(), nullptr, 0, &w, &h) != RTP_OK) { ++bad; continue; }
    out.resize((size_t)w * h * 3);
    if (rtp_decode_image(b.data(), b.size(), out.data(), out.size(), &w, &h) == RTP_OK) ++ok; else ++bad;
  }
  printf("asan fuzz: %ld decoded, %ld rejected\n", ok, bad);
  return 0;
}

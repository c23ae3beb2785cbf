// Mutation fuzzer for csrc/netdef.cpp (prototxt text parser, binary caffemodel reader) under ASan + UBSan.  From the repo root:
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 tools/fuzz_netdef.cpp -o /tmp/fuzz_netdef && ASAN_OPTIONS=detect_leaks=0 /tmp/fuzz_netdef SEED ITERS
#include "../caffe_rtpose_amd/csrc/netdef.cpp"
#include <random>
using namespace rtp;
int main(int argc, char** argv) {
  std::mt19937 rng(argc > 1 ? atoi(argv[1]) : 1);
  const int N = argc > 2 ? atoi(argv[2]) : 2000;
  // prototxt
  const std::string proto = emit_prototxt(build_linevec(0));
  long ok = 0, bad = 0;
  for (int it = 0; it < N; ++it) {
    std::string t = proto;
    const int mode = rng() % 4;
    if (mode == 0) for (int k = 0, m = 1 + rng() % 6; k < m; ++k) t[rng() % t.size()] = (char)(32 + rng() % 95);
    else if (mode == 1) t.resize(1 + rng() % t.size());
    else if (mode == 2) { size_t i = rng() % t.size(); t.insert(i, std::string(1 + rng() % 8, "{}:\"#\n"[rng() % 6])); }
    else { size_t i = rng() % t.size(), n = rng() % 200; t.erase(i, n); }
    NetDef nd; std::string err;
    if (parse_prototxt(t, &nd, &err)) ++ok; else ++bad;
  }
  printf("prototxt fuzz: %ld parsed, %ld rejected\n", ok, bad);
  // caffemodel: a small synthetic file
  std::vector<LayerWeights> lw;
  for (int l = 0; l < 3; ++l) {
    LayerWeights L; L.name = "conv" + std::to_string(l); L.type = "Convolution";
    BlobData w, b; w.shape = {4, 3, 3, 3}; w.data.assign(108, 0.5f); b.shape = {4}; b.data.assign(4, 0.1f);
    L.blobs = {w, b}; lw.push_back(L);
  }
  std::string err;
  write_caffemodel("/tmp/fz.caffemodel", "net", lw, &err);
  std::ifstream f("/tmp/fz.caffemodel", std::ios::binary);
  std::string base((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  ok = bad = 0;
  for (int it = 0; it < N; ++it) {
    std::string t = base;
    const int mode = rng() % 3;
    if (mode == 0) for (int k = 0, m = 1 + rng() % 6; k < m; ++k) t[rng() % t.size()] = (char)rng();
    else if (mode == 1) t.resize(1 + rng() % t.size());
    else { size_t i = rng() % t.size(); t.insert(i, std::string(1 + rng() % 6, (char)rng())); }
    { std::ofstream o("/tmp/fz2.caffemodel", std::ios::binary); o.write(t.data(), t.size()); }
    std::vector<LayerWeights> out;
    if (read_caffemodel("/tmp/fz2.caffemodel", &out, &err)) ++ok; else ++bad;
  }
  printf("caffemodel fuzz: %ld parsed, %ld rejected\n", ok, bad);
  return 0;
}

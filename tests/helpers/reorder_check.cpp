// reorder_check.cpp — CPU test of rtpose.bin's re-orderer (row a12: buffer_and_order, rtpose.cpp:1214-1273):
// frames leave in index order, indices registered as dropped are skipped without waiting, and when more than
// BUFFER_SIZE (4) frames wait behind a gap the smallest is force-emitted.  Includes the CLI's translation unit.
#define main rtpose_cli_main
#include "../../caffe_rtpose_amd/csrc/rtpose_main.cpp"
#undef main

static std::vector<int> run_case(const std::vector<int>& arrivals, const std::vector<int>& dropped) {
  {
    std::lock_guard<std::mutex> l(G.mutex);
    while (!G.dropped_index.empty()) G.dropped_index.pop();
    for (int d : dropped) G.dropped_index.push(d);
  }
  std::atomic<bool> done{false};
  std::thread t(reorderer, &done);
  for (int idx : arrivals) {
    Frame f;
    f.index = idx;
    G.output_queue.push(std::move(f));
    std::this_thread::sleep_for(std::chrono::milliseconds(2));  // let the re-orderer see the frames one at a time
  }
  done = true;
  t.join();
  std::vector<int> out;
  Frame f;
  while (G.output_queue_ordered.try_pop(&f)) out.push_back(f.index);
  return out;
}

static int expect(const char* name, const std::vector<int>& got, const std::vector<int>& want) {
  if (got == want) { printf("ok   %s\n", name); return 0; }
  printf("FAIL %s: got", name);
  for (int v : got) printf(" %d", v);
  printf(" want");
  for (int v : want) printf(" %d", v);
  printf("\n");
  return 1;
}

int main() {
  int bad = 0;
  bad += expect("scrambled arrival", run_case({2, 1, 3, 5, 4}, {}), {1, 2, 3, 4, 5});
  bad += expect("dropped index is skipped", run_case({1, 2, 4, 5}, {3}), {1, 2, 4, 5});
  bad += expect("first frames dropped", run_case({3, 4}, {1, 2}), {3, 4});
  bad += expect("window overflow forces the smallest out", run_case({2, 3, 4, 5, 6, 7}, {}), {2, 3, 4, 5, 6, 7});
  bad += expect("late frame after a forced emit still leaves", run_case({2, 3, 4, 5, 6, 1}, {}), {2, 3, 4, 5, 6, 1});
  return bad;
}

// Launch cost of a small kernel as a function of its private (scratch) segment
// size: above some total size the runtime allocates and frees the scratch per
// dispatch.  Prints microseconds per launch for several per-lane sizes.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WORDS>
__global__ void k(int* out, int n) {
  int buf[WORDS];
  for (int i = 0; i < WORDS; i++) buf[i] = i * n;
  int s = 0;
  for (int i = 0; i < n; i++) s += buf[(i * 7 + threadIdx.x) % WORDS];
  out[threadIdx.x] = s;
}

template <int WORDS>
void run(int* out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; i++) k<WORDS><<<1, 64>>>(out, 5);
  hipDeviceSynchronize();
  hipEventRecord(a);
  const int reps = 50;
  for (int i = 0; i < reps; i++) k<WORDS><<<1, 64>>>(out, 5);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("scratch %5d B/lane: %8.1f us per launch\n", WORDS * 4, 1000.0 * ms / reps);
}

__global__ void plain(int* out) { out[threadIdx.x] = threadIdx.x; }

template <int WORDS>
void runAlternating(int* out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; i++) { k<WORDS><<<1, 64>>>(out, 5); plain<<<1, 64>>>(out); }
  hipDeviceSynchronize();
  hipEventRecord(a);
  const int reps = 50;
  for (int i = 0; i < reps; i++) { k<WORDS><<<1, 64>>>(out, 5); plain<<<1, 64>>>(out); }
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("scratch %5d B/lane alternating with a scratch-free kernel: %8.1f us per pair\n", WORDS * 4, 1000.0 * ms / reps);
  // with a host synchronisation between the two, like a level loop
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) { k<WORDS><<<1, 64>>>(out, 5); hipStreamSynchronize(0); plain<<<1, 64>>>(out); hipStreamSynchronize(0); }
  hipEventRecord(b);
  hipEventSynchronize(b);
  hipEventElapsedTime(&ms, a, b);
  printf("scratch %5d B/lane, synchronised after every launch:        %8.1f us per pair\n", WORDS * 4, 1000.0 * ms / reps);
}

int main() {
  int* out;
  hipMalloc(&out, 4096);
  run<16>(out);
  run<64>(out);
  run<69>(out);
  run<70>(out);
  run<96>(out);
  run<196>(out);
  run<512>(out);
  run<64>(out);
  runAlternating<64>(out);
  runAlternating<196>(out);
  runAlternating<512>(out);
  return 0;
}

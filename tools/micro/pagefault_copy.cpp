// How fast can N host threads copy into FRESH (never touched) pageable memory?  (the tail of a host-pointer linscan call:
// 80 MB of results into the caller's new arrays).  g++ -O2 -pthread tools/micro/pagefault_copy.cpp -o /tmp/pfc && /tmp/pfc
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
int main() {
  const size_t bytes = (size_t)80 << 20;
  char *src = (char *)malloc(bytes);
  memset(src, 1, bytes);
  for (int nt : {1, 2, 4, 8, 16}) {
    for (int rep = 0; rep < 3; ++rep) {
      char *dst = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      const size_t per = bytes / nt;
      for (int t = 0; t < nt; ++t) th.emplace_back([=] { memcpy(dst + t * per, src + t * per, per); });
      for (auto &t : th) t.join();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rep == 2) printf("threads=%2d  fresh pages: %.2f ms  %.1f GB/s\n", nt, ms, bytes / ms / 1e6);
      auto t1 = std::chrono::steady_clock::now();
      memcpy(dst, src, bytes);
      const double ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
      if (rep == 2 && nt == 1) printf("            touched pages, 1 thread: %.2f ms  %.1f GB/s\n", ms2, bytes / ms2 / 1e6);
      munmap(dst, bytes);
    }
  }
  return 0;
}

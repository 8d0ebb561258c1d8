// Development probe for the file-to-file route: T threads pread N files of a list into one buffer -- pageable (malloc) or
// page-locked (hipHostMalloc) -- each file opened per read or through fds opened beforehand. usage: io_read list.txt
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static void pfor(size_t n, unsigned nt, F f) {
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) th.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
  for (auto &t : th) t.join();
}
int main(int argc, char **argv) {
  std::vector<std::string> files;
  std::ifstream in(argv[1]);
  for (std::string s; std::getline(in, s);) if (!s.empty()) files.push_back(s);
  const size_t n = files.size(), per = 320044;
  char *pageable = (char *)malloc(n * per), *pinned = nullptr;
  memset(pageable, 1, n * per);
  double t0 = now();
  if (hipHostMalloc((void **)&pinned, n * per, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
  printf("hipHostMalloc %.2f GB: %.3f s\n", n * per / 1e9, now() - t0);
  for (unsigned nt : {8u, 16u, 32u, 64u, 128u}) {
    for (int mode = 0; mode < 3; ++mode) {                 // 0: pageable, open per file; 1: pinned, open per file; 2: pinned, fds opened before
      char *dst = mode == 0 ? pageable : pinned;
      std::vector<int> fds(n, -1);
      double t_open = 0;
      if (mode == 2) { t0 = now(); pfor(n, nt, [&](size_t i) { fds[i] = open(files[i].c_str(), O_RDONLY); }); t_open = now() - t0; }
      t0 = now();
      pfor(n, nt, [&](size_t i) {
        const int fd = mode == 2 ? fds[i] : open(files[i].c_str(), O_RDONLY);
        size_t got = 0;
        while (got < per) { const ssize_t k = pread(fd, dst + i * per + got, per - got, got); if (k <= 0) break; got += k; }
        close(fd);
      });
      const double dt = now() - t0;
      printf("threads %3u mode %d: read %.3f s = %.1f GB/s%s\n", nt, mode, dt, n * per / dt / 1e9, mode == 2 ? (" (opens " + std::to_string(t_open) + " s)").c_str() : "");
    }
  }
  return 0;
}

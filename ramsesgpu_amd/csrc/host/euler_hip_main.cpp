// euler_hip -- command line front end, the counterpart of the reference's euler_cpu / euler_gpu
// (src/euler_main.cpp:76-193): euler_hip --param file.ini [--set "section.key=value;..."]
#include <cstdio>
#include <cstring>
#include <string>

#include "../../../include/rgpu.h"

int main(int argc, char** argv) {
  std::string param, overrides;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--param") && i + 1 < argc) param = argv[++i];
    else if (!std::strcmp(argv[i], "--set") && i + 1 < argc) overrides = argv[++i];
    else if (!std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h")) {
      std::printf("usage: %s --param <file.ini> [--set \"section.key=value;...\"]\n", argv[0]);
      return 0;
    }
  }
  if (param.empty()) {
    std::fprintf(stderr, "missing --param <file.ini>\n");
    return 2;
  }
  char err[512] = {0};
  double mcell = 0.0;
  const int n = rgpuh_run(param.c_str(), overrides.c_str(), &mcell, err, sizeof(err));
  if (n < 0) {
    std::fprintf(stderr, "euler_hip: %s\n", err);
    return 1;
  }
  std::printf("steps %d  %.3f Mcell-updates/s\n", n, mcell);
  return 0;
}

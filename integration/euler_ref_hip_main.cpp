// euler_ref_hip_main.cpp -- the reference's run classes driven through the MI355X binding: what src/euler_main.cpp:130-200
// does for the unsplit Godunov scheme, with MHDRunGodunovHip / HydroRunGodunovHip in place of MHDRunGodunov /
// HydroRunGodunov.  Everything but the hot path is reference code (ConfigMap, init_simulation, start(), outputVtk);
// linked from the reference's own objects by oracle/Makefile.ref into oracle/_ref/euler_ref_hip (test infrastructure:
// tests/test_integration_stub.py compares its .vti output with euler_cpu's golden fixtures on the GPU box).
//   usage: euler_ref_hip --param file.ini
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>

#include "HydroRunGodunovHip.h"
#include "MHDRunGodunovHip.h"

int main(int argc, char* argv[]) {
  std::string input;
  for (int a = 1; a + 1 < argc; ++a)
    if (!std::strcmp(argv[a], "--param")) input = argv[a + 1];
  if (input.empty()) { std::fprintf(stderr, "usage: %s --param file.ini\n", argv[0]); return 2; }
  ConfigMap configMap(input);
  const bool mhd = configMap.getBool("MHD", "enable", false);
  try {
    hydroSimu::HydroRunBase* run = mhd ? static_cast<hydroSimu::HydroRunBase*>(new hydroSimu::MHDRunGodunovHip(configMap))
                                       : static_cast<hydroSimu::HydroRunBase*>(new hydroSimu::HydroRunGodunovHip(configMap));
    std::cout << "backend : " << rgpu_backend_name() << std::endl;
    run->start();
    delete run;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "euler_ref_hip: %s\n", e.what());
    return 1;
  }
  return 0;
}

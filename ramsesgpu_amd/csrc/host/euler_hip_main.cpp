// euler_hip -- command line front end, the counterpart of the reference's euler_cpu / euler_gpu
// (src/euler_main.cpp:76-193): euler_hip --param file.ini [--set "section.key=value;..."]
// and of euler_mpi_main.cpp:76 for z-slabs, one process per GPU:
//   for r in 0..N-1:  RANK=r WORLD_SIZE=N LOCAL_RANK=r euler_hip --slabs N --param file.ini [--rendezvous /tmp/file] &
// (or any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK, e.g. python -m torch.distributed.run --no-python).  Rank 0
// creates the RCCL unique id and publishes it through the rendezvous file; the halo exchange and the 1/dt all-reduce run
// over RCCL (include/rgpu_comm.h).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "../../../include/rgpu_comm.h"

namespace {
int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }

// rank 0 writes the id (write + rename: readers never see a partial file), the others poll for it
bool rendezvous(const std::string& path, int rank, char id[RGPU_COMM_ID_BYTES]) {
  if (rank == 0) {
    if (rgpu_comm_unique_id(id)) return false;
    const std::string tmp = path + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(id, 1, RGPU_COMM_ID_BYTES, f) == RGPU_COMM_ID_BYTES;
    std::fclose(f);
    return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
  }
  for (int tries = 0; tries < 6000; ++tries) {   // up to 60 s
    FILE* f = std::fopen(path.c_str(), "rb");
    if (f) {
      const size_t n = std::fread(id, 1, RGPU_COMM_ID_BYTES, f);
      std::fclose(f);
      if (n == RGPU_COMM_ID_BYTES) return true;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  return false;
}
}  // namespace

int main(int argc, char** argv) {
  std::string param, overrides, rdv;
  int slabs = 0;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--param") && i + 1 < argc) param = argv[++i];
    else if (!std::strcmp(argv[i], "--set") && i + 1 < argc) overrides = argv[++i];
    else if (!std::strcmp(argv[i], "--slabs") && i + 1 < argc) slabs = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--rendezvous") && i + 1 < argc) rdv = argv[++i];
    else if (!std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h")) {
      std::printf("usage: %s --param <file.ini> [--set \"section.key=value;...\"] [--slabs N [--rendezvous <file>]]\n"
                  "  --slabs N: this process is rank $RANK of $WORLD_SIZE (= N) z-slabs and drives GPU $LOCAL_RANK\n", argv[0]);
      return 0;
    }
  }
  if (param.empty()) {
    std::fprintf(stderr, "missing --param <file.ini>\n");
    return 2;
  }
  char err[512] = {0};
  double mcell = 0.0;
  int n;
  if (slabs > 0) {
    // RCCL's peer buffers between the rank processes: this pool's host driver supports dmabuf IPC only (set before the first HIP call)
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", slabs), local = env_int("LOCAL_RANK", rank);
    if (world != slabs || rank < 0 || rank >= world) {
      std::fprintf(stderr, "euler_hip: --slabs %d but RANK=%d WORLD_SIZE=%d\n", slabs, rank, world);
      return 2;
    }
    if (rdv.empty()) rdv = std::string("/tmp/rgpu_rendezvous_") + (std::getenv("MASTER_PORT") ? std::getenv("MASTER_PORT") : "0");
    char id[RGPU_COMM_ID_BYTES];
    if (!rendezvous(rdv, rank, id)) {
      std::fprintf(stderr, "euler_hip: rendezvous through %s failed\n", rdv.c_str());
      return 1;
    }
    n = rgpuh_run_slabs(param.c_str(), overrides.c_str(), rank, world, local, id, &mcell, err, sizeof(err));
    if (rank == 0) std::remove(rdv.c_str());
    if (n >= 0 && rank != 0) return 0;
  } else {
    n = rgpuh_run(param.c_str(), overrides.c_str(), &mcell, err, sizeof(err));
  }
  if (n < 0) {
    std::fprintf(stderr, "euler_hip: %s\n", err);
    return 1;
  }
  std::printf("steps %d  %.3f Mcell-updates/s\n", n, mcell);
  return 0;
}

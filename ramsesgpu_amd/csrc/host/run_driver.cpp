#include "run_driver.h"
#include "hdf5_io.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "init_conditions.h"

namespace rgpu_host {

GodunovRun::GodunovRun(const IniConfig& cfg, int slab_rank, int slab_count)
    : cfg_(cfg), ctx_(0), totalTime_(0.0), restart_has_ghosts_(false), warned_no_hdf5_(false), wrote_hdf5_(false), hooked_(false),
      noted_vtk_(false), noted_hist_(false) {
  std::memset(&hooks_, 0, sizeof(hooks_));
  params_from_config(cfg_, slab_rank, slab_count, &p_, &rs_);
  const int rc = rgpu_create(&p_, &ctx_);
  if (rc) {
    const std::string msg = ctx_ ? rgpu_last_error(ctx_) : "allocation failure";
    if (ctx_) rgpu_destroy(ctx_);
    ctx_ = 0;
    throw std::runtime_error("rgpu_create: " + msg);
  }
  h_U_.assign(rgpu_state_elems(&p_), 0.0);
}

GodunovRun::~GodunovRun() {
  if (ctx_) rgpu_destroy(ctx_);
}

void GodunovRun::check(int rc, const char* what) {
  if (rc) throw std::runtime_error(std::string(what) + ": " + rgpu_last_error(ctx_));
}

int GodunovRun::init_simulation() {
  int timeStep = 0;
  if (rs_.restartEnabled) {
    // restart run (HydroRunBase.cpp:7033-7066): the state comes from a file of an earlier run instead of the problem's
    // initial condition; the static gravity / forcing fields below are rebuilt as in a fresh run.  The reference reads
    // HDF5 (inputHdf5, :4818-5160): so does this driver for a .h5 name (hdf5_io.h, libhdf5 bound at run time); it also resumes
    // from its raw .rgr dump and from the .vti it writes (interior cells, raw doubles, step count and time in the header).
    std::fill(h_U_.begin(), h_U_.end(), 0.0);
    const std::string path = rs_.outputDir + "/" + rs_.restartFilename;
    restart_has_ghosts_ = false;
    const bool h5 = path.size() > 3 && path.substr(path.size() - 3) == ".h5";
    const bool dump = h5 || (path.size() > 4 && path.substr(path.size() - 4) == ".rgr");
    if (rs_.restartUpscale && slab()) throw std::runtime_error("restart_upscale is single-domain only");
    if (rs_.restartUpscale) {
      if (!dump) throw std::runtime_error("restart_upscale reads an .h5 file or a .rgr dump ([output] outputHdf5=yes in the coarse run)");
      if (p_.nx % 2 || p_.ny % 2 || (p_.nz_global != 1 && p_.nz % 2)) throw std::runtime_error("restart_upscale: nx, ny, nz must be even");
      inputRestartUpscaled(path, &restart_has_ghosts_);   // timeStep stays 0; the forcing process starts afresh (:7084-7088)
    } else {
      timeStep = dump ? inputRestart(path, &restart_has_ghosts_) : inputVtk(path);
      restore_forcing_process(timeStep);
    }
    std::cout << "### This is a restarted run ! Current time is " << totalTime_ << " (step " << timeStep << ") ###\n";
  } else {
    init_condition(cfg_, p_, h_U_.data());
  }
  check(rgpu_upload(ctx_, h_U_.data(), 1), "upload");
  if (p_.gravityEnabled == 2) {   // h_gravity of the problem, copied to the device once (d_gravity.copyFromHost)
    std::vector<double> hG(3 * (h_U_.size() / p_.nbVar));
    if (init_gravity_field(cfg_, p_, hG.data())) check(rgpu_set_gravity_field(ctx_, hG.data()), "set_gravity_field");
  }
  if (p_.randomForcingEnabled) {   // h_randomForcing -> d_randomForcing (HydroRunBase.cpp:7199-7209)
    std::vector<double> hF(3 * (h_U_.size() / p_.nbVar));
    if (init_forcing_field(cfg_, p_, hF.data())) check(rgpu_set_forcing_field(ctx_, hF.data()), "set_forcing_field");
  }
  return timeStep;
}

// collective outcome of a step that can fail on one rank alone: every rank throws if any rank failed
void GodunovRun::agree_or_throw(const std::string& local_error, const char* what) {
  if (!hooked_) { if (!local_error.empty()) throw std::runtime_error(local_error); return; }
  if (!hooks_.agree) {   // an older slab driver: barrier only (a failing rank then throws alone)
    if (!local_error.empty()) throw std::runtime_error(local_error);
    hook_check(hooks_.barrier(hooks_.self), "barrier");
    return;
  }
  const int failed = hooks_.agree(hooks_.self, local_error.empty() ? 0 : 1);
  if (failed < 0) hook_check(failed, what);
  if (failed > 0) {
    std::ostringstream m;
    m << what << ": " << failed << " of " << p_.slab_count << " ranks failed";
    if (!local_error.empty()) m << "; this rank (" << p_.slab_rank << "): " << local_error;
    throw std::runtime_error(m.str());
  }
}

void GodunovRun::hook_check(int rc, const char* what) {
  if (rc) throw std::runtime_error(std::string(what) + ": " + (hooks_.last_error ? hooks_.last_error(hooks_.self) : "slab driver error"));
}
void GodunovRun::note_once(bool* flag, const char* msg) {
  if (!*flag && p_.slab_rank == 0) std::cerr << msg << "\n";
  *flag = true;
}

void GodunovRun::make_all_boundaries(int parity) {
  if (hooked_) { hook_check(hooks_.make_all_boundaries(hooks_.self, parity, totalTime_, 0.0), "make_all_boundaries"); return; }
  check(rgpu_make_all_boundaries(ctx_, parity, totalTime_, 0.0), "make_all_boundaries");
}

double GodunovRun::compute_dt(int useU) {
  if (hooked_) { double d = 0.0; hook_check(hooks_.compute_dt(hooks_.self, useU, &d), "compute_dt"); return d; }
  const double dt = rgpu_compute_dt(ctx_, useU);
  if (!(dt == dt)) throw std::runtime_error(std::string("compute_dt: ") + rgpu_last_error(ctx_));
  return dt;
}

void GodunovRun::godunov_unsplit(int nStep, double dt) { check(rgpu_godunov_unsplit(ctx_, nStep, dt, totalTime_), "godunov_unsplit"); }

// MHDRunGodunov.cpp:4077-4089 / HydroRunGodunov.cpp:4082-4126 (unsplit branch)
void GodunovRun::oneStepIntegration(int& nStep, double& t, double& dt) {
  if (hooked_) { hook_check(hooks_.one_step_integration(hooks_.self, &nStep, &t, &dt), "oneStepIntegration"); return; }
  dt = compute_dt(nStep % 2);
  godunov_unsplit(nStep, dt);
  nStep++;
  t += dt;
}

void GodunovRun::copyGpuToCpu(int nStep) { check(rgpu_download(ctx_, h_U_.data(), nStep % 2), "download"); }

// Hand-written VTI writer of the reference (HydroRunBase.cpp:2681-2995): ImageData, PointData, appended raw
// little-endian Float64, one uint32 byte count per array, INTERIOR cells only, names density..bz -- the reference's bytes.
void GodunovRun::outputVtk(int nStep) {
  static const char* names[8] = {"density", "energy", "mx", "my", "mz", "bx", "by", "bz"};
  const int gw = p_.ghostWidth, nx = p_.nx, ny = p_.ny;
  const bool three_d = p_.nz_global != 1;
  const int nz = three_d ? p_.nz : 1;
  const size_t isize = nx + 2 * gw, jsize = ny + 2 * gw, ksize = three_d ? nz + 2 * gw : 1;
  const size_t ncell = isize * jsize * ksize;
  std::ostringstream fn;
  fn << rs_.outputDir << "/" << rs_.outputPrefix << "_" << std::setw(7) << std::setfill('0') << nStep << ".vti";
  std::ofstream out(fn.str().c_str(), std::ios::binary);
  if (!out) throw std::runtime_error("cannot write " + fn.str());
  if (rs_.outputVtkAscii) {   // [output] outputVtkAscii=yes: the reference's text form, byte for byte (HydroRunBase.cpp:2919-2973)
    out << "<?xml version=\"1.0\"?>\n<VTKFile type=\"ImageData\" version=\"0.1\" byte_order=\"LittleEndian\">\n";
    out << "  <ImageData WholeExtent=\"0 " << nx - 1 << " 0 " << ny - 1 << " 0 " << nz - 1 << "\" Origin=\"0 0 0\" Spacing=\"1 1 1\">\n";
    out << "  <Piece Extent=\"0 " << nx - 1 << " 0 " << ny - 1 << " 0 " << nz - 1 << "\">\n    <PointData>\n";
    for (int v = 0; v < p_.nbVar; ++v) {
      const char* nm = (p_.nbVar == 4 && v == 3) ? "my" : names[v];
      out << "      <DataArray type=\"Float64\" Name=\"" << nm << "\" format=\"ascii\" >\n";
      for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j) {
          const size_t kk = three_d ? k + gw : 0;
          const double* src = &h_U_[(gw) + isize * ((j + gw) + jsize * kk) + ncell * v];
          for (int i = 0; i < nx; ++i) out << std::setprecision(12) << src[i] << " ";
          out << "\n";
        }
      out << "      </DataArray>\n";
    }
    out << "    </PointData>\n    <CellData>\n    </CellData>\n  </Piece>\n  </ImageData>\n</VTKFile>\n";
    return;
  }
  // appended raw doubles: the bytes of the reference's file (no XML declaration in this mode), followed by ONE comment line
  // with what a restart needs besides the fields (the reference keeps them as HDF5 attributes "time step" / "total time")
  // the format's per-array byte count is a uint32 (the reference's files: HydroRunBase.cpp:2974-2995): >= 4 GiB per variable cannot
  // be described -- say so instead of writing a corrupt header (HDF5 output has no such limit)
  if (sizeof(double) * (size_t)nx * ny * nz > UINT32_MAX)
    throw std::runtime_error("outputVtk: " + std::to_string(sizeof(double) * (size_t)nx * ny * nz) + " bytes per variable exceed the uint32 byte count of the "
                             "appended-raw .vti format; use [output] outputHdf5=yes (or outputVtkAscii=yes) for this box");
  const uint32_t nbytes = static_cast<uint32_t>(sizeof(double) * nx * ny * nz);
  out << "<VTKFile type=\"ImageData\" version=\"0.1\" byte_order=\"LittleEndian\">\n";
  out << "  <ImageData WholeExtent=\"0 " << nx - 1 << " 0 " << ny - 1 << " 0 " << nz - 1 << "\" Origin=\"0 0 0\" Spacing=\"1 1 1\">\n";
  out << "  <Piece Extent=\"0 " << nx - 1 << " 0 " << ny - 1 << " 0 " << nz - 1 << "\">\n    <PointData>\n";
  for (int v = 0; v < p_.nbVar; ++v) {
    const char* nm = (p_.nbVar == 4 && v == 3) ? "my" : names[v];
    out << "     <DataArray type=\"Float64\" Name=\"" << nm << "\" format=\"appended\" offset=\""
        << static_cast<size_t>(v) * (nbytes + sizeof(uint32_t)) << "\" />\n";
  }
  out << "    </PointData>\n    <CellData>\n    </CellData>\n  </Piece>\n  </ImageData>\n  <AppendedData encoding=\"raw\">\n_";
  std::vector<double> line(nx);
  for (int v = 0; v < p_.nbVar; ++v) {
    out.write(reinterpret_cast<const char*>(&nbytes), sizeof(nbytes));
    for (int k = 0; k < nz; ++k)
      for (int j = 0; j < ny; ++j) {
        const size_t kk = three_d ? k + gw : 0;
        const double* src = &h_U_[(gw) + isize * ((j + gw) + jsize * kk) + ncell * v];
        out.write(reinterpret_cast<const char*>(src), sizeof(double) * nx);
      }
  }
  out << "  </AppendedData>\n</VTKFile>\n";
  char tail[160];
  std::snprintf(tail, sizeof(tail), "<!-- rgpu restart: nStep=%d totalTime=%a (%.17g) -->\n", nStep, totalTime_, totalTime_);
  out << tail;
}

// z-slab runs: HydroRunBaseMpi::outputVtk, hand-written branch (HydroRunBaseMpi.cpp:4167-4790) for the 1 x 1 x N topology of the
// slab driver -- every rank writes <prefix>_time<step>_mpi<rank>.vti with its own planes (ranks > 0: plus the plane below, the
// one-cell overlap the parallel image-data format wants), rank 0 the index <prefix>_time<step>.pvti.  Appended raw doubles or,
// [output] outputVtkAscii, text with the stream's default precision (the MPI writer sets none).
void GodunovRun::outputVtkSlab(int nStep) {
  static const char* names[8] = {"density", "energy", "mx", "my", "mz", "bx", "by", "bz"};
  const int gw = p_.ghostWidth, nx = p_.nx, ny = p_.ny, nz = p_.nz, rank = p_.slab_rank, count = p_.slab_count;
  const size_t isize = nx + 2 * gw, jsize = ny + 2 * gw, ksize = nz + 2 * gw;
  const size_t ncell = isize * jsize * ksize;
  std::ostringstream ts, rk;
  ts << std::setw(7) << std::setfill('0') << nStep;
  rk << std::setw(5) << std::setfill('0') << rank;
  const std::string stem = rs_.outputPrefix + "_time" + ts.str();
  std::string local_error;
  try {
    if (rank == 0) {
      const std::string hn = rs_.outputDir + "/" + stem + ".pvti";
      std::ofstream h(hn.c_str());
      if (!h) throw std::runtime_error("cannot write " + hn);
      h << "<?xml version=\"1.0\"?>\n<VTKFile type=\"PImageData\" version=\"0.1\" byte_order=\"LittleEndian\">\n";
      h << "  <PImageData WholeExtent=\"0 " << nx - 1 << " 0 " << ny - 1 << " 0 " << count * nz - 1
        << "\" GhostLevel=\"0\" Origin=\"0 0 0\" Spacing=\"1 1 1\">\n    <PPointData Scalars=\"Scalars_\">\n";
      for (int v = 0; v < p_.nbVar; ++v) h << "      <PDataArray type=\"Float64\" Name=\"" << names[v] << "\"/>\n";
      h << "    </PPointData>\n";
      for (int r = 0; r < count; ++r) {
        std::ostringstream pr;
        pr << std::setw(5) << std::setfill('0') << r;
        h << " <Piece Extent=\"0 " << nx - 1 << " 0 " << ny - 1 << " ";
        if (r == 0) h << 0 << " " << nz - 1 << " ";
        else h << r * nz - 1 << " " << r * nz + nz - 1 << " ";
        h << "\" Source=\"" << stem << "_mpi" << pr.str() << ".vti\"/>\n";
      }
      h << "</PImageData>\n</VTKFile>\n";
    }
    const std::string fn = rs_.outputDir + "/" + stem + "_mpi" + rk.str() + ".vti";
    std::ofstream out(fn.c_str(), std::ios::binary);
    if (!out) throw std::runtime_error("cannot write " + fn);
    const int zlo = (rank == 0) ? 0 : rank * nz - 1, zhi = (rank == 0) ? nz - 1 : rank * nz + nz - 1;
    const int k0 = (rank == 0) ? gw : gw - 1, k1 = (int)ksize - gw;   // array planes [k0, k1)
    if (rs_.outputVtkAscii) out << "<?xml version=\"1.0\"?>\n";
    out << "<VTKFile type=\"ImageData\" version=\"0.1\" byte_order=\"LittleEndian\">\n";
    out << "  <ImageData WholeExtent=\"0 " << nx - 1 << " 0 " << ny - 1 << " " << zlo << " " << zhi << "\" Origin=\"0 0 0\" Spacing=\"1 1 1\">\n";
    out << "  <Piece Extent=\"0 " << nx - 1 << " 0 " << ny - 1 << " " << zlo << " " << zhi << "\">\n    <PointData>\n";
    if (rs_.outputVtkAscii) {
      for (int v = 0; v < p_.nbVar; ++v) {
        out << "      <DataArray type=\"Float64\" Name=\"" << names[v] << "\" format=\"ascii\">\n";
        for (int k = k0; k < k1; ++k)
          for (int j = gw; j < (int)jsize - gw; ++j) {
            const double* src = &h_U_[gw + isize * (j + jsize * (size_t)k) + ncell * v];
            for (int i = 0; i < nx; ++i) out << src[i] << " ";
            out << "\n";
          }
        out << "      </DataArray>\n";
      }
      out << "    </PointData>\n    <CellData>\n    </CellData>\n  </Piece>\n  </ImageData>\n</VTKFile>\n";
    } else {
      const size_t tuples = (size_t)nx * ny * (size_t)(k1 - k0);
      if (tuples * sizeof(double) > UINT32_MAX)   // caught below: every rank then fails the output step together (hooks.agree)
        throw std::runtime_error("outputVtk: " + std::to_string(tuples * sizeof(double)) + " bytes per variable in this rank's piece exceed the uint32 byte "
                                 "count of the appended-raw .vti format; use [output] outputHdf5=yes for this box");
      const uint32_t nbytes = static_cast<uint32_t>(tuples * sizeof(double));
      for (int v = 0; v < p_.nbVar; ++v)
        out << "     <DataArray type=\"Float64\" Name=\"" << names[v] << "\" format=\"appended\" offset=\""
            << (size_t)v * tuples * sizeof(double) + (size_t)v * sizeof(uint32_t) << "\" />\n";
      out << "    </PointData>\n    <CellData>\n    </CellData>\n  </Piece>\n  </ImageData>\n  <AppendedData encoding=\"raw\">\n_";
      for (int v = 0; v < p_.nbVar; ++v) {
        out.write(reinterpret_cast<const char*>(&nbytes), sizeof(nbytes));
        for (int k = k0; k < k1; ++k)
          for (int j = gw; j < (int)jsize - gw; ++j)
            out.write(reinterpret_cast<const char*>(&h_U_[gw + isize * (j + jsize * (size_t)k) + ncell * v]), sizeof(double) * nx);
      }
      out << "  </AppendedData>\n</VTKFile>\n";
    }
    if (!out) throw std::runtime_error("writing " + fn + " failed");
  } catch (const std::exception& e) { local_error = e.what(); }
  agree_or_throw(local_error, "outputVtk");
}

// Xsmurf: one ASCII header line + the interior of ONE variable (the density: the reference's default argument) as raw doubles,
// written to the current directory (no outputDir in the name, HydroRunBase.cpp:2535)
void GodunovRun::outputXsm(int nStep) {
  const int gw = p_.ghostWidth, nx = p_.nx, ny = p_.ny;
  const bool three_d = p_.nz_global != 1;
  const int nz = three_d ? p_.nz : 1;
  const size_t isize = nx + 2 * gw, jsize = ny + 2 * gw;
  std::ostringstream fn;
  fn << rs_.outputPrefix << "_d_" << std::setw(7) << std::setfill('0') << nStep << ".xsm";
  std::ofstream out(fn.str().c_str(), std::ios::binary);
  if (!out) throw std::runtime_error("cannot write " + fn.str());
  if (three_d) out << "Binary 1 " << nx << "x" << ny << "x" << nz << " " << nx * ny * nz << "(" << sizeof(double) << " byte reals)\n";
  else out << "Binary 1 " << nx << "x" << ny << " " << nx * ny << "(" << sizeof(double) << " byte reals)\n";
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j) {
      const size_t kk = three_d ? k + gw : 0;
      out.write(reinterpret_cast<const char*>(&h_U_[gw + isize * ((j + gw) + jsize * kk)]), sizeof(double) * nx);
    }
}

// NRRD: a text header + the interior of each variable converted to 32-bit floats, one file per variable (d, p, u, v, w, a, b, c)
void GodunovRun::outputNrrd(int nStep) {
  static const char* prefix[8] = {"d", "p", "u", "v", "w", "a", "b", "c"};
  const int gw = p_.ghostWidth, nx = p_.nx, ny = p_.ny;
  const bool three_d = p_.nz_global != 1;
  const int nz = three_d ? p_.nz : 1;
  const size_t isize = nx + 2 * gw, jsize = ny + 2 * gw, ksize = three_d ? nz + 2 * gw : 1, ncell = isize * jsize * ksize;
  std::vector<float> row(nx);
  for (int v = 0; v < p_.nbVar; ++v) {
    const int pv = (p_.nbVar == 4 && v == 3) ? 3 : v;   // 2D hydro: d, p, u, v
    std::ostringstream fn;
    fn << rs_.outputDir << "/" << rs_.outputPrefix << "_" << prefix[pv] << "_" << std::setw(7) << std::setfill('0') << nStep << ".nrrd";
    std::ofstream out(fn.str().c_str(), std::ios::binary);
    if (!out) throw std::runtime_error("cannot write " + fn.str());
    out << "NRRD0004\n# Complete NRRD file format specification at:\n# http://teem.sourceforge.net/nrrd/format.html\ntype: float\n";
    if (three_d) out << "dimension: 3\nsizes: " << nx << " " << ny << " " << nz << "\nspace directions: (1,0,0) (0,1,0) (0,0,1)\n";
    else out << "dimension: 2\nsizes: " << nx << " " << ny << "\nspace directions: (1,0) (0,1)\n";
    out << "endian: little\nencoding: raw\n\n";
    for (int k = 0; k < nz; ++k)
      for (int j = 0; j < ny; ++j) {
        const size_t kk = three_d ? k + gw : 0;
        const double* src = &h_U_[gw + isize * ((j + gw) + jsize * kk) + ncell * v];
        for (int i = 0; i < nx; ++i) row[i] = (float)src[i];
        out.write(reinterpret_cast<const char*>(row.data()), sizeof(float) * nx);
      }
  }
}

H5Box GodunovRun::h5_box(int nx, int ny, int nz) const {
  H5Box b;
  b.nx = nx; b.ny = ny; b.nz = nz;
  b.ghostWidth = p_.ghostWidth; b.nbVar = p_.nbVar;
  b.three_d = p_.nz_global != 1; b.mhd = p_.mhdEnabled != 0;
  return b;
}

// [output] outputHdf5=yes: the reference's HDF5 file (hdf5_io.h) when libhdf5 can be loaded, the raw dump below otherwise
void GodunovRun::outputHdf5(int nStep) {
  const char* fmt = std::getenv("RGPU_RESTART_FORMAT");   // "rgr": the raw dump even when HDF5 is there
  std::string why;
  if (slab()) {   // (agreed by all ranks: libhdf5 may be missing on one node only)
    std::string local_error;
    if ((fmt && std::string(fmt) == "rgr") || !hdf5_available(&why))
      local_error = "outputs of a z-slab run go to one HDF5 file for the whole box: " + (why.empty() ? std::string("RGPU_RESTART_FORMAT=rgr is single-domain only") : why);
    agree_or_throw(local_error, "outputHdf5");
  }
  if (fmt && std::string(fmt) == "rgr") { outputRestart(nStep); return; }
  if (!hdf5_available(&why)) {
    if (!warned_no_hdf5_) { std::cerr << "outputHdf5: " << why << " -- writing raw .rgr dumps instead\n"; warned_no_hdf5_ = true; }
    outputRestart(nStep);
    return;
  }
  std::ostringstream fn;
  fn << rs_.outputDir << "/" << rs_.outputPrefix << "_" << std::setw(7) << std::setfill('0') << nStep << ".h5";
  const bool three_d = p_.nz_global != 1;
  if (slab()) {   // one file for the whole box: the slabs take turns, rank 0 creates it (hdf5_io.h)
    // A write can fail on ONE rank (disk full, file lock, libhdf5 missing on that node): the ranks agree on the outcome of
    // every turn (hooks.agree: a sum all-reduce of the failure flags in place of the bare barrier) and throw TOGETHER --
    // a rank that left this loop alone would leave the others waiting in the collective for ever.
    for (int r = 0; r < p_.slab_count; ++r) {
      std::string local_error;
      if (r == p_.slab_rank) {
        try {
          hdf5_write_slab(fn.str(), h_U_.data(), h5_box(p_.nx, p_.ny, p_.nz), p_.nz_global, p_.slab_rank, p_.slab_count, r == 0, rs_.ghostIncluded, nStep,
                          totalTime_, rs_.hdf5CompressionLevel);
        } catch (const std::exception& e) { local_error = e.what(); }
      }
      agree_or_throw(local_error, "outputHdf5");
    }
  } else {
    hdf5_write_state(fn.str(), h_U_.data(), h5_box(p_.nx, p_.ny, three_d ? p_.nz : 1), rs_.ghostIncluded, nStep, totalTime_, rs_.hdf5CompressionLevel);
  }
  wrote_hdf5_ = true;
}

// Raw restart dump: one text line "RGPU-RESTART 1 nx ny nz ghostWidth nbVar ghostIncluded nStep totalTime(hex float)",
// then nbVar arrays of little-endian doubles, x fastest -- the whole ghost-inclusive arrays when ghostIncluded (what a
// shearing-box run needs: the field on the first high x face is evolved by the CT update and not rebuilt by the ghost
// fill), the interior otherwise.  Stands in for outputHdf5 / inputHdf5 (HydroRunBase.cpp:3308-3640, 4818-5160).
void GodunovRun::outputRestart(int nStep) {
  const int gw = p_.ghostWidth, nx = p_.nx, ny = p_.ny;
  const bool three_d = p_.nz_global != 1;
  const int nz = three_d ? p_.nz : 1;
  const size_t isize = nx + 2 * gw, jsize = ny + 2 * gw, ksize = three_d ? nz + 2 * gw : 1;
  const size_t ncell = isize * jsize * ksize;
  std::ostringstream fn;
  fn << rs_.outputDir << "/" << rs_.outputPrefix << "_" << std::setw(7) << std::setfill('0') << nStep << ".rgr";
  std::ofstream out(fn.str().c_str(), std::ios::binary);
  if (!out) throw std::runtime_error("cannot write " + fn.str());
  char line[256];
  std::snprintf(line, sizeof(line), "RGPU-RESTART 1 %d %d %d %d %d %d %d %a\n", nx, ny, nz, gw, p_.nbVar, rs_.ghostIncluded ? 1 : 0, nStep, totalTime_);
  out << line;
  if (rs_.ghostIncluded) {
    out.write(reinterpret_cast<const char*>(h_U_.data()), sizeof(double) * ncell * p_.nbVar);
  } else {
    for (int v = 0; v < p_.nbVar; ++v)
      for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j) {
          const size_t kk = three_d ? k + gw : 0;
          out.write(reinterpret_cast<const char*>(&h_U_[gw + isize * ((j + gw) + jsize * kk) + ncell * v]), sizeof(double) * nx);
        }
  }
}

// Reads a dump of an nx x ny x nz box (ghost width and variables of this run) into dst, a ghost-inclusive array of that
// box; without ghosts in the file the ghost cells of dst are left as they are.
int GodunovRun::read_restart(const std::string& path, int nx_want, int ny_want, int nz_want, double* dst, bool* ghosts_read) {
  if (path.size() > 3 && path.substr(path.size() - 3) == ".h5") {   // inputHdf5 (HydroRunBase.cpp:4818-5155)
    double t = 0.0;
    const int step = slab() ? hdf5_read_slab(path, dst, h5_box(nx_want, ny_want, nz_want), p_.nz_global, p_.slab_rank, p_.slab_count, &t, ghosts_read)
                            : hdf5_read_state(path, dst, h5_box(nx_want, ny_want, nz_want), &t, ghosts_read);
    totalTime_ = rs_.restartResetTotalTime ? 0.0 : t;
    return step;
  }
  if (slab()) throw std::runtime_error("restart: a z-slab run resumes from the .h5 file of the whole box");
  std::ifstream in(path.c_str(), std::ios::binary);
  if (!in) throw std::runtime_error("restart: cannot read " + path);
  std::string line;
  std::getline(in, line);
  int ver = 0, nx = 0, ny = 0, nz = 0, gw = 0, nv = 0, gi = 0, nStep = 0;
  char hex[64] = {0};
  if (std::sscanf(line.c_str(), "RGPU-RESTART %d %d %d %d %d %d %d %d %63s", &ver, &nx, &ny, &nz, &gw, &nv, &gi, &nStep, hex) != 9 || ver != 1)
    throw std::runtime_error("restart: " + path + " is not a restart dump of this code");
  const bool three_d = p_.nz_global != 1;
  if (nx != nx_want || ny != ny_want || nz != nz_want || gw != p_.ghostWidth || nv != p_.nbVar)
    throw std::runtime_error("restart: " + path + " holds another box than expected from [mesh] nx, ny, nz / other variables");
  const size_t isize = nx + 2 * gw, jsize = ny + 2 * gw, ksize = three_d ? nz + 2 * gw : 1;
  const size_t ncell = isize * jsize * ksize;
  if (gi) {
    in.read(reinterpret_cast<char*>(dst), sizeof(double) * ncell * nv);
  } else {
    for (int v = 0; v < nv; ++v)
      for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j) {
          const size_t kk = three_d ? k + gw : 0;
          in.read(reinterpret_cast<char*>(&dst[gw + isize * ((j + gw) + jsize * kk) + ncell * v]), sizeof(double) * nx);
        }
  }
  if (!in) throw std::runtime_error("restart: " + path + " is truncated");
  if (ghosts_read) *ghosts_read = gi != 0;
  totalTime_ = rs_.restartResetTotalTime ? 0.0 : std::strtod(hex, 0);
  return nStep;
}

int GodunovRun::inputRestart(const std::string& path, bool* ghosts_read) {
  const bool three_d = p_.nz_global != 1;
  return read_restart(path, p_.nx, p_.ny, three_d ? p_.nz : 1, h_U_.data(), ghosts_read);
}

// [run] restart_upscale: the dump holds the same problem on nx/2 x ny/2 (x nz/2) cells; every cell of this run's arrays,
// ghosts included, takes the value of the coarse cell under it -- (index + ghostWidth) / 2 per direction -- except the
// face-centred field: a fine face lying on a coarse face takes that face's value, one in the middle of a coarse cell the
// mean of the two coarse faces around it along the component's own direction, which keeps div B = 0
// (HydroRunBase::upscale, HydroRunBase.cpp:5170-5278; call site :7047-7062).  In 2D Bz is cell-like.  The step count of
// the coarse run is not taken over (the reference drops inputHdf5's return value there), its time is.
void GodunovRun::inputRestartUpscaled(const std::string& path, bool* ghosts_read) {
  const bool three_d = p_.nz_global != 1;
  const int gw = p_.ghostWidth, nv = p_.nbVar;
  const int lnx = p_.nx / 2, lny = p_.ny / 2, lnz = three_d ? p_.nz / 2 : 1;
  const size_t li = lnx + 2 * gw, lj = lny + 2 * gw, lk = three_d ? lnz + 2 * gw : 1, lcell = li * lj * lk;
  std::vector<double> low(lcell * nv, 0.0);
  read_restart(path, lnx, lny, lnz, low.data(), ghosts_read);
  const size_t isize = p_.nx + 2 * gw, jsize = p_.ny + 2 * gw, ksize = three_d ? p_.nz + 2 * gw : 1, ncell = isize * jsize * ksize;
  const bool mhd = nv == 8;
  const int ncopy = mhd ? 5 : nv;   // cell-centred variables (2D MHD: density, energy and the three momenta)
  for (size_t k = 0; k < ksize; ++k) {
    const size_t kl = three_d ? (k + gw) / 2 : 0;
    for (size_t j = 0; j < jsize; ++j) {
      const size_t jl = (j + gw) / 2;
      for (size_t i = 0; i < isize; ++i) {
        const size_t il = (i + gw) / 2;
        const size_t hi = i + isize * (j + jsize * k), lo = il + li * (jl + lj * kl);
        for (int v = 0; v < ncopy; ++v) h_U_[hi + ncell * v] = low[lo + lcell * v];
        if (mhd) {
          const double* bx = &low[lcell * 5]; const double* by = &low[lcell * 6]; const double* bz = &low[lcell * 7];
          h_U_[hi + ncell * 5] = (i + gw == 2 * il) ? bx[lo] : (bx[lo] + bx[lo + 1]) / 2;
          h_U_[hi + ncell * 6] = (j + gw == 2 * jl) ? by[lo] : (by[lo] + by[lo + li]) / 2;
          if (!three_d) h_U_[hi + ncell * 7] = bz[lo];
          else h_U_[hi + ncell * 7] = (k + gw == 2 * kl) ? bz[lo] : (bz[lo] + bz[lo + li * lj]) / 2;
        }
      }
    }
  }
}

// Reads a .vti written by outputVtk (same box, same variables) into the interior of h_U_; returns the step count of the
// file and sets totalTime_ (0 / 0 for files without the restart comment).  Ghost cells stay zero: start() fills them.
int GodunovRun::inputVtk(const std::string& path) {
  std::ifstream in(path.c_str(), std::ios::binary);
  if (!in) throw std::runtime_error("restart: cannot read " + path);
  std::string blob((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  const size_t mark = blob.find("<AppendedData encoding=\"raw\">");
  if (mark == std::string::npos) throw std::runtime_error("restart: " + path + " is not an appended-raw .vti");
  const std::string head = blob.substr(0, mark);
  int nStep = 0;
  double t = 0.0;
  {
    // the comment line after </VTKFile> (files of earlier versions carried it in the header)
    const std::string tail = blob.substr(blob.size() > 256 ? blob.size() - 256 : 0);
    const size_t ct = tail.find("rgpu restart: nStep="), ch = head.find("rgpu restart: nStep=");
    const char* at = ct != std::string::npos ? tail.c_str() + ct : ch != std::string::npos ? head.c_str() + ch : 0;
    if (at) {
      char hex[64] = {0};
      if (std::sscanf(at, "rgpu restart: nStep=%d totalTime=%63s", &nStep, hex) == 2) t = std::strtod(hex, 0);
    }
  }
  int e[6] = {0, 0, 0, 0, 0, 0};
  {
    const size_t w = head.find("WholeExtent=\"");
    if (w == std::string::npos || std::sscanf(head.c_str() + w, "WholeExtent=\"%d %d %d %d %d %d\"", &e[0], &e[1], &e[2], &e[3], &e[4], &e[5]) != 6)
      throw std::runtime_error("restart: no WholeExtent in " + path);
  }
  const int gw = p_.ghostWidth, nx = p_.nx, ny = p_.ny;
  const bool three_d = p_.nz_global != 1;
  const int nz = three_d ? p_.nz : 1;
  if (e[1] - e[0] + 1 != nx || e[3] - e[2] + 1 != ny || e[5] - e[4] + 1 != nz)
    throw std::runtime_error("restart: " + path + " holds another box than [mesh] nx, ny, nz");
  const size_t isize = nx + 2 * gw, jsize = ny + 2 * gw, ksize = three_d ? nz + 2 * gw : 1;
  const size_t ncell = isize * jsize * ksize;
  const size_t start = blob.find('_', mark) + 1;
  const size_t nbytes = sizeof(double) * nx * ny * nz;
  size_t pos = 0;
  int nvar = 0;
  while ((pos = head.find("format=\"appended\" offset=\"", pos)) != std::string::npos) {
    pos += std::strlen("format=\"appended\" offset=\"");
    const size_t off = std::strtoull(head.c_str() + pos, 0, 10);
    if (nvar >= p_.nbVar) throw std::runtime_error("restart: " + path + " holds more variables than this run");
    uint32_t count = 0;
    if (start + off + sizeof(count) + nbytes > blob.size()) throw std::runtime_error("restart: " + path + " is truncated");
    std::memcpy(&count, blob.data() + start + off, sizeof(count));
    if (count != nbytes) throw std::runtime_error("restart: unexpected array size in " + path);
    const char* src = blob.data() + start + off + sizeof(count);
    for (int k = 0; k < nz; ++k)
      for (int j = 0; j < ny; ++j) {
        const size_t kk = three_d ? k + gw : 0;
        std::memcpy(&h_U_[gw + isize * ((j + gw) + jsize * kk) + ncell * nvar], src + sizeof(double) * nx * (j + (size_t)ny * k), sizeof(double) * nx);
      }
    ++nvar;
  }
  if (nvar != p_.nbVar) throw std::runtime_error("restart: " + path + " holds fewer variables than this run");
  totalTime_ = rs_.restartResetTotalTime ? 0.0 : t;
  return nStep;
}

// the forcing process of a step, next to the fields (the reference: <prefix>_forcing_NNNNNNN.npz, output_forcing)
void GodunovRun::save_forcing_process(int nStep) {
  if (p_.ouForcingEnabled) {
    double st[RGPU_OU_STATE_DOUBLES];
    check(rgpu_ou_forcing_get_state(ctx_, st), "ou_forcing_get_state");
    std::ostringstream fo;
    fo << rs_.outputDir << "/" << rs_.outputPrefix << "_forcing_" << std::setw(7) << std::setfill('0') << nStep << ".txt";
    std::ofstream f(fo.str().c_str());
    if (!f) throw std::runtime_error("cannot write " + fo.str());
    char buf[64];
    for (int i = 0; i < RGPU_OU_STATE_DOUBLES; ++i) { std::snprintf(buf, sizeof(buf), "%a\n", st[i]); f << buf; }
  }
}

// the Ornstein-Uhlenbeck process of step nStep, if the earlier run wrote it (outputVtk / outputRestart do)
void GodunovRun::restore_forcing_process(int nStep) {
  if (p_.ouForcingEnabled) {   // the forcing process of that step, if the earlier run wrote it
    std::ostringstream fo;
    fo << rs_.outputDir << "/" << rs_.outputPrefix << "_forcing_" << std::setw(7) << std::setfill('0') << nStep << ".txt";
    std::ifstream f(fo.str().c_str());
    if (f) {
      double st[RGPU_OU_STATE_DOUBLES];
      std::string tok;
      int n = 0;
      while (n < RGPU_OU_STATE_DOUBLES && (f >> tok)) st[n++] = std::strtod(tok.c_str(), 0);
      if (n != RGPU_OU_STATE_DOUBLES) throw std::runtime_error("restart: " + fo.str() + " is incomplete");
      check(rgpu_ou_forcing_set_state(ctx_, st), "ou_forcing_set_state");
    } else {
      std::cerr << "restart: no " << fo.str() << ", the forcing process starts afresh\n";
    }
  }
}

// MHDRunGodunov.cpp:3801-4070 / HydroRunGodunov.cpp:3857-4080 (no restart, no history, VTI outputs only)
// history(nStep, dt): the history file of the MHD runs (MHDRunBase.cpp:3285-3619), one row per call, same columns and
// ostream formatting: MRI problems get history_mri's eleven columns, Orszag-Tang history_default's four, every other
// problem none (history_empty); the turbulence problems get history_turbulence's twenty columns.  The sums come from the device
// (rgpu_history_mri) instead of a copy of the state to the host.  The inertial-wave problem gets history_inertial_wave's
// probe row (:3414-3469): the velocity of one cell in units of cIso, read with rgpu_read_cell.
void GodunovRun::history(int nStep, double dt) {
  if (!p_.mhdEnabled) return;
  const std::string problem = cfg_.get_string("hydro", "problem", "unknown");
  const bool mri = problem == "MRI" || problem == "Mri" || problem == "mri";
  bool dflt = problem == "Orszag-Tang" || problem == "OrszagTang";
  if (slab()) {
    // z-slab runs follow the history set-up of the MPI classes (HydroRunBaseMpi.cpp:10667-10730): MRI -> history_mhd_mri,
    // turbulence -> history_mhd_turbulence (16 columns, no DFT amplitudes), every other MHD problem -> history_mhd_default
    const bool turb = (problem == "turbulence" || problem == "turbulence-Ornstein-Uhlenbeck") && p_.nz_global != 1;
    if (turb && hooks_.history_turbulence) {
      double h[14];
      hook_check(hooks_.history_turbulence(hooks_.self, nStep % 2, h), "history");   // collective; rank 0's values are the row
      if (p_.slab_rank != 0) return;
      const std::string fileName = cfg_.get_string("output", "outputDir", "./") + "/" + cfg_.get_string("output", "outputPrefix", "output") +
                                   "_" + cfg_.get_string("history", "filename", "history.txt");
      std::ofstream histo(fileName.c_str(), std::ios::out | std::ios::app | std::ios::ate);
      if (totalTime_ <= 0) {
        histo << "# history" << std::endl;
        histo << "# totalTime dt mass divB eKin eMag helicity mean_B mean_Bx mean_By mean_Bz mean_rhovx mean_rhovy mean_rhovz Ma_s Ma_alfven\n";
      }
      histo << totalTime_ << "\t" << dt;
      for (int q = 0; q < 14; ++q) histo << "\t" << h[q];
      histo << "\n";
      return;
    }
    if (!mri) dflt = true;
  }
  const bool iwave = problem == "InertialWave" || problem == "inertialwave" || problem == "Inertial-Wave" ||
                     problem == "inertial-wave" || problem == "Inertialwave";
  if (iwave && !slab()) {
    double u[8];
    const int iPos = p_.ghostWidth + p_.nx / 2, kPos = p_.ghostWidth;
    if (p_.nz_global == 1) check(rgpu_read_cell(ctx_, nStep % 2, iPos, kPos, 0, u), "history");
    else check(rgpu_read_cell(ctx_, nStep % 2, iPos, 1, kPos, u), "history");
    const std::string fileName = cfg_.get_string("output", "outputDir", "./") + "/" + cfg_.get_string("output", "outputPrefix", "output") +
                                 "_" + cfg_.get_string("history", "filename", "history.txt");
    std::ofstream histo(fileName.c_str(), std::ios::out | std::ios::app | std::ios::ate);
    if (totalTime_ <= 0) histo << "# history" << std::endl;
    const double rho = u[RGPU_ID], dvx = u[RGPU_IU] / rho, dvy = u[RGPU_IV] / rho;
    // the reference's row, missing separator between totalTime and dt included; fmt() = " " + setw(12) fixed, 8 digits
    histo << totalTime_ << "" << dt << " " << rho << " ";
    const std::ios_base::fmtflags flags = histo.flags();
    histo << " " << std::setw(12) << std::setprecision(8) << std::fixed << dvx / p_.cIso;
    histo.flags(flags);
    histo << " ";
    histo << " " << std::setw(12) << std::setprecision(8) << std::fixed << dvy / p_.cIso;
    histo.flags(flags);
    histo << "\n";
    return;
  }
  if (!slab() && (problem == "turbulence" || problem == "turbulence-Ornstein-Uhlenbeck")) {   // history_turbulence (MHDRunBase.cpp:3626-3810)
    if (p_.nz_global == 1) return;
    double h[18];
    check(rgpu_history_turbulence(ctx_, nStep % 2, h), "history");
    const std::string fileName = cfg_.get_string("output", "outputDir", "./") + "/" + cfg_.get_string("output", "outputPrefix", "output") +
                                 "_" + cfg_.get_string("history", "filename", "history.txt");
    std::ofstream histo(fileName.c_str(), std::ios::out | std::ios::app | std::ios::ate);
    if (totalTime_ <= 0) {
      histo << "# history" << std::endl;
      histo << "# totalTime dt mass divB eKin eMag helicity mean_rho mean_B mean_Bx mean_By mean_Bz mean_rhovx mean_rhovy mean_rhovz Ma_s Ma_alfven coef_x coef_y coef_z\n";
    }
    histo << totalTime_ << "\t" << dt;
    for (int q = 0; q < 18; ++q) histo << "\t" << h[q];
    histo << "\n";
    return;
  }
  if (!mri && !dflt) return;
  if (mri && p_.nz_global == 1) return;   // history_mri does nothing in 2D
  double h[8];
  if (slab()) hook_check(hooks_.history_mri(hooks_.self, nStep % 2, h), "history");   // the same sums on every rank; rank 0 writes
  else check(rgpu_history_mri(ctx_, nStep % 2, h), "history");
  if (p_.slab_rank != 0) return;
  const std::string fileName = cfg_.get_string("output", "outputDir", "./") + "/" + cfg_.get_string("output", "outputPrefix", "output") +
                               "_" + cfg_.get_string("history", "filename", "history.txt");
  std::ofstream histo(fileName.c_str(), std::ios::out | std::ios::app | std::ios::ate);
  if (totalTime_ <= 0) {
    histo << "# history" << std::endl;
    if (mri) histo << "# totalTime dt mass maxwell reynolds maxwell+reynolds magp mean_Bx mean_By mean_Bz divB\n";
    else histo << "# totalTime dt mass divB\n";
  }
  if (mri)
    histo << totalTime_ << "\t" << dt << "\t" << h[0] << "\t" << h[1] << "\t" << h[2] << "\t" << h[1] + h[2] << "\t" << h[3] << "\t"
          << h[4] << "\t" << h[5] << "\t" << h[6] << "\t" << h[7] << "\n";
  else
    histo << totalTime_ << "\t" << dt << "\t" << h[0] << "\t" << h[7] << "\n";
}

int GodunovRun::start(double* mcell_per_s, rgpuh_attach_fn attach, void* user) {
  int nStep = init_simulation();
  if (attach) {   // the slab driver takes over the stepping from here (rgpuh_run_hooked)
    rgpuh_step_hooks h;
    std::memset(&h, 0, sizeof(h));
    if (attach(user, ctx_, &h) || !h.make_all_boundaries || !h.compute_dt || !h.one_step_integration || !h.barrier)
      throw std::runtime_error(std::string("attaching the slab driver failed") + (h.last_error ? std::string(": ") + h.last_error(h.self) : std::string()));
    set_hooks(h);
  }
  if (slab() && !hooked_) throw std::runtime_error("a z-slab context needs the slab driver (rgpuh_run_hooked / euler_hip --slabs)");
  if (!(rs_.restartEnabled && restart_has_ghosts_)) {   // a restart file with ghosts needs no fill (MHDRunGodunov.cpp:3818-3824)
    make_all_boundaries(0);
    // h_U.copyTo(h_U2): refresh both device arrays from the ghost-filled one
    copyGpuToCpu(0);
    check(rgpu_upload(ctx_, h_U_.data(), 1), "upload");
  }
  if (!rs_.restartEnabled) totalTime_ = 0.0;   // a restarted run keeps the time of its file (MHDRunGodunov.cpp:3866-3880)
  double dt = compute_dt(0);
  if (p_.slab_rank == 0) std::cout << "Initial dt : " << std::setprecision(8) << dt << std::endl;
  double io_seconds = 0.0;
  // history cadence of MHDRunGodunov::start (MHDRunGodunov.cpp:3913-3916, 3975-3984)
  const bool historyEnabled = p_.mhdEnabled && cfg_.get_bool("history", "enabled", false);
  const double dtHist = cfg_.get_float("history", "dtHist", static_cast<float>(10 * dt));
  double tHist = totalTime_;   // MHDRunGodunov.cpp:3916
  // z-slab runs: the MRI / Orszag-Tang history goes through the slab driver's global sums; the other problems' do not
  const bool slab_history_ok = slab() && hooks_.history_mri;   // (every MHD problem has a history in the MPI classes)
  const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  while (totalTime_ < rs_.tEnd && nStep < rs_.nStepmax) {
    if (rs_.nLog > 0 && (nStep % rs_.nLog) == 0 && p_.slab_rank == 0)
      std::printf("  step=%9d t=%14.8f dt=%16.12f\n", nStep, totalTime_, dt);
    if (rs_.nOutput > 0 && (nStep % rs_.nOutput) == 0) {   // noutput <= 0: no output (the reference divides by zero here)
      const std::chrono::steady_clock::time_point w0 = std::chrono::steady_clock::now();
      const bool raw = (rs_.outputXsm || rs_.outputNrrd) && !slab();
      if (rs_.outputVtk || rs_.outputRestart || raw) copyGpuToCpu(nStep);
      if (raw && rs_.outputXsm) outputXsm(nStep);
      if (raw && rs_.outputNrrd) outputNrrd(nStep);
      if (rs_.outputVtk && !slab()) outputVtk(nStep);
      if (rs_.outputVtk && slab()) outputVtkSlab(nStep);
      if (rs_.outputRestart) outputHdf5(nStep);
      if ((rs_.outputVtk || rs_.outputRestart) && p_.slab_rank == 0) save_forcing_process(nStep);
      io_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
      if (p_.slab_rank == 0) std::printf("  step=%9d t=%14.8f dt=%16.12f\n", nStep, totalTime_, dt);
    }
    if (historyEnabled && slab() && !slab_history_ok) note_once(&noted_hist_, "z-slab run: this problem's history file is not written");
    if (historyEnabled && (!slab() || slab_history_ok) && (tHist == 0 || ((totalTime_ - dt <= tHist + dtHist) && (totalTime_ > tHist + dtHist)))) {
      history(nStep, dt);
      tHist += dtHist;
    }
    // the turns of this loop up to the next one that logs, writes or records history do nothing but step: rgpu_run_steps runs them
    // (same states, same dt sequence; where a step is one fused kernel the time step stays on the device in between)
    int quiet = 1;
    if ((!hooked_ || hooks_.run_steps) && !historyEnabled) {
      quiet = rs_.nStepmax - nStep;
      if (rs_.nLog > 0 && rs_.nLog - nStep % rs_.nLog < quiet) quiet = rs_.nLog - nStep % rs_.nLog;
      if (rs_.nOutput > 0 && rs_.nOutput - nStep % rs_.nOutput < quiet) quiet = rs_.nOutput - nStep % rs_.nOutput;
    }
    if (quiet > 1) {
      if (hooked_) {
        const int rc = hooks_.run_steps(hooks_.self, quiet, rs_.tEnd, &nStep, &totalTime_, &dt);
        if (rc < 0) hook_check(rc, "run_steps");
      } else {
        const int rc = rgpu_run_steps(ctx_, quiet, rs_.tEnd, &nStep, &totalTime_, &dt);
        if (rc < 0) check(rc, "run_steps");
      }
    } else {
      oneStepIntegration(nStep, totalTime_, dt);
    }
  }
  check(rgpu_synchronize(ctx_), "synchronize");
  if (hooked_) hook_check(hooks_.barrier(hooks_.self), "barrier");
  {
    const std::chrono::steady_clock::time_point w0 = std::chrono::steady_clock::now();
    const bool raw = (rs_.outputXsm || rs_.outputNrrd) && !slab();
    if (rs_.outputVtk || rs_.outputRestart || raw) copyGpuToCpu(nStep);
    if (raw && rs_.outputXsm) outputXsm(nStep);
    if (raw && rs_.outputNrrd) outputNrrd(nStep);
    if (rs_.outputVtk && !slab()) outputVtk(nStep);
    if (rs_.outputVtk && slab()) outputVtkSlab(nStep);
    if (rs_.outputRestart) outputHdf5(nStep);
    if ((rs_.outputVtk || rs_.outputRestart) && p_.slab_rank == 0) save_forcing_process(nStep);
    // the XDMF index of the .h5 files of this run, in the current directory (MHDRunGodunov.cpp:4004)
    if (wrote_hdf5_ && p_.slab_rank == 0)
      xdmf_write_wrapper(rs_.outputPrefix, h5_box(p_.nx, p_.ny, (p_.nz_global != 1) ? p_.nz_global : 1), rs_.ghostIncluded, nStep, rs_.nOutput);
    io_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
  }
  const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double nz = (p_.nz_global != 1) ? p_.nz_global : 1;   // a slab reports the rate of the whole box
  const double rate = 1.0 * nStep * p_.nx * p_.ny * nz / (total - io_seconds);
  if (p_.slab_rank == 0) {
    std::cout << "DEBUG : totalTime " << std::setprecision(12) << totalTime_ << std::endl;
    std::cout << "####################################\nGlobal performance                  \n"
              << rate << " cell updates per seconds (based on wall time)\n####################################\n";
  }
  if (mcell_per_s) *mcell_per_s = rate / 1e6;
  return nStep;
}

}  // namespace rgpu_host

extern "C" int rgpuh_run_hooked(const char* ini_path, const char* overrides, int slab_rank, int slab_count, rgpuh_attach_fn attach,
                                rgpuh_detach_fn detach, void* user, double* mcell_per_s, char* err, int err_len) {
  int n = RGPU_EINVAL;
  try {
    if (!attach) throw std::runtime_error("run_hooked: no attach function");
    rgpu_host::IniConfig cfg;
    const int rc = cfg.load_file(ini_path ? ini_path : "");
    if (rc != 0) throw std::runtime_error(std::string("cannot read parameter file ") + (ini_path ? ini_path : "(null)"));
    if (overrides) cfg.apply_overrides(overrides);
    rgpu_host::GodunovRun run(cfg, slab_rank, slab_count);
    try {
      n = run.start(mcell_per_s, attach, user);
    } catch (...) {
      if (detach) detach(user);   // the slab driver refers to the context: release it before the run object goes
      throw;
    }
    if (detach) detach(user);
  } catch (const std::exception& e) {
    if (err && err_len > 0) std::snprintf(err, static_cast<size_t>(err_len), "%s", e.what());
    return RGPU_EINVAL;
  }
  return n;
}

extern "C" int rgpuh_run(const char* ini_path, const char* overrides, double* mcell_per_s, char* err, int err_len) {
  try {
    rgpu_host::IniConfig cfg;
    const int rc = cfg.load_file(ini_path ? ini_path : "");
    if (rc != 0) throw std::runtime_error(std::string("cannot read parameter file ") + (ini_path ? ini_path : "(null)"));
    if (overrides) cfg.apply_overrides(overrides);
    rgpu_host::GodunovRun run(cfg);
    return run.start(mcell_per_s);
  } catch (const std::exception& e) {
    if (err && err_len > 0) std::snprintf(err, static_cast<size_t>(err_len), "%s", e.what());
    return RGPU_EINVAL;
  }
}

// hdf5_io.cpp -- see hdf5_io.h.  libhdf5 (>= 1.10) is loaded with dlopen and called through the handful of entry points below.
#include "hdf5_io.h"

#include <dlfcn.h>

#include <cstdint>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <stdexcept>
#include <utility>
#include <vector>

namespace rgpu_host {

namespace {

// the part of the HDF5 1.10 / 1.12 / 1.14 C API used here (hid_t is 64-bit since 1.10)
typedef int64_t hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;
const unsigned kAccRdonly = 0x0000u, kAccRdwr = 0x0001u, kAccTrunc = 0x0002u;   // H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC
const hid_t kDefault = 0;                                   // H5P_DEFAULT
const int kScalar = 0, kSelectSet = 0, kScopeLocal = 0;     // H5S_SCALAR, H5S_SELECT_SET, H5F_SCOPE_LOCAL
const size_t kVariable = (size_t)-1;                        // H5T_VARIABLE

struct Api {
  void* so;
  herr_t (*open)();
  herr_t (*get_libversion)(unsigned*, unsigned*, unsigned*);
  herr_t (*Eset_auto2)(hid_t, void*, void*);
  hid_t (*Fcreate)(const char*, unsigned, hid_t, hid_t);
  hid_t (*Fopen)(const char*, unsigned, hid_t);
  herr_t (*Fflush)(hid_t, int);
  herr_t (*Fclose)(hid_t);
  hid_t (*Screate_simple)(int, const hsize_t*, const hsize_t*);
  hid_t (*Screate)(int);
  herr_t (*Sselect_hyperslab)(hid_t, int, const hsize_t*, const hsize_t*, const hsize_t*, const hsize_t*);
  int (*Sget_simple_extent_ndims)(hid_t);
  int (*Sget_simple_extent_dims)(hid_t, hsize_t*, hsize_t*);
  herr_t (*Sclose)(hid_t);
  hid_t (*Pcreate)(hid_t);
  herr_t (*Pset_chunk)(hid_t, int, const hsize_t*);
  herr_t (*Pset_shuffle)(hid_t);
  herr_t (*Pset_deflate)(hid_t, unsigned);
  herr_t (*Pclose)(hid_t);
  hid_t (*Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t);
  hid_t (*Dopen2)(hid_t, const char*, hid_t);
  hid_t (*Dget_space)(hid_t);
  herr_t (*Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*);
  herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*);
  herr_t (*Dclose)(hid_t);
  hid_t (*Acreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t);
  hid_t (*Aopen)(hid_t, const char*, hid_t);
  herr_t (*Awrite)(hid_t, hid_t, const void*);
  herr_t (*Aread)(hid_t, hid_t, void*);
  herr_t (*Aclose)(hid_t);
  hid_t (*Tcopy)(hid_t);
  herr_t (*Tset_size)(hid_t, size_t);
  herr_t (*Tclose)(hid_t);
  hid_t native_double, native_int, c_s1, cls_dataset_create;   // values of the library's global ids (valid after H5open)
};

Api g_api;
int g_state = 0;   // 0 = not tried, 1 = loaded, -1 = unavailable
std::string g_why;

template <class F>
bool sym(void* so, const char* name, F* out) {
  *out = reinterpret_cast<F>(dlsym(so, name));
  if (!*out) g_why = std::string("libhdf5 lacks ") + name;
  return *out != 0;
}
bool global_id(void* so, const char* name, hid_t* out) {
  const hid_t* p = reinterpret_cast<const hid_t*>(dlsym(so, name));
  if (!p) { g_why = std::string("libhdf5 lacks ") + name; return false; }
  *out = *p;
  return true;
}

// binds one candidate library; false (g_why set) if it cannot be loaded, lacks an entry point or is older than 1.10
bool bind(const std::string& name) {
  void* so = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!so) { g_why = name + " cannot be loaded"; return false; }
  Api a;
  a.so = so;
  const bool ok =
      sym(so, "H5open", &a.open) && sym(so, "H5get_libversion", &a.get_libversion) && sym(so, "H5Eset_auto2", &a.Eset_auto2) &&
      sym(so, "H5Fcreate", &a.Fcreate) && sym(so, "H5Fopen", &a.Fopen) && sym(so, "H5Fflush", &a.Fflush) && sym(so, "H5Fclose", &a.Fclose) &&
      sym(so, "H5Screate_simple", &a.Screate_simple) && sym(so, "H5Screate", &a.Screate) && sym(so, "H5Sselect_hyperslab", &a.Sselect_hyperslab) &&
      sym(so, "H5Sget_simple_extent_ndims", &a.Sget_simple_extent_ndims) && sym(so, "H5Sget_simple_extent_dims", &a.Sget_simple_extent_dims) &&
      sym(so, "H5Sclose", &a.Sclose) && sym(so, "H5Pcreate", &a.Pcreate) && sym(so, "H5Pset_chunk", &a.Pset_chunk) &&
      sym(so, "H5Pset_shuffle", &a.Pset_shuffle) && sym(so, "H5Pset_deflate", &a.Pset_deflate) && sym(so, "H5Pclose", &a.Pclose) &&
      sym(so, "H5Dcreate2", &a.Dcreate2) && sym(so, "H5Dopen2", &a.Dopen2) && sym(so, "H5Dget_space", &a.Dget_space) &&
      sym(so, "H5Dwrite", &a.Dwrite) && sym(so, "H5Dread", &a.Dread) && sym(so, "H5Dclose", &a.Dclose) && sym(so, "H5Acreate2", &a.Acreate2) &&
      sym(so, "H5Aopen", &a.Aopen) && sym(so, "H5Awrite", &a.Awrite) && sym(so, "H5Aread", &a.Aread) && sym(so, "H5Aclose", &a.Aclose) &&
      sym(so, "H5Tcopy", &a.Tcopy) && sym(so, "H5Tset_size", &a.Tset_size) && sym(so, "H5Tclose", &a.Tclose);
  unsigned maj = 0, min = 0, rel = 0;
  if (ok && (a.open() < 0 || a.get_libversion(&maj, &min, &rel) < 0)) { g_why = name + ": H5open failed"; dlclose(so); return false; }
  if (ok && maj == 1 && min < 10) { g_why = name + " is older than HDF5 1.10 (32-bit handles)"; dlclose(so); return false; }
  if (!ok || !global_id(so, "H5T_NATIVE_DOUBLE_g", &a.native_double) || !global_id(so, "H5T_NATIVE_INT_g", &a.native_int) ||
      !global_id(so, "H5T_C_S1_g", &a.c_s1) || !global_id(so, "H5P_CLS_DATASET_CREATE_ID_g", &a.cls_dataset_create)) {
    g_why = name + ": " + g_why;
    dlclose(so);
    return false;
  }
  a.Eset_auto2(0, 0, 0);   // no error stack dumps on stderr: failures are reported through exceptions here
  g_api = a;
  return true;
}

bool load() {
  if (g_state) return g_state > 0;
  g_state = -1;
  std::vector<std::string> names;
  if (const char* e = std::getenv("RGPU_HDF5_LIB")) names.push_back(e);   // "none": no HDF5 (the callers fall back to the raw dump)
  else {
    const char* defaults[] = {"libhdf5.so", "libhdf5.so.103", "libhdf5_serial.so", "libhdf5_serial.so.103", "libhdf5.so.200", "libhdf5.so.310",
                              "/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so"};
    for (const char* d : defaults) names.push_back(d);
  }
  std::string reasons;
  for (const std::string& n : names) {
    if (bind(n)) { g_state = 1; return true; }
    reasons += (reasons.empty() ? "" : "; ") + g_why;
  }
  g_why = "no usable HDF5 library (" + reasons + "; set RGPU_HDF5_LIB)";
  return false;
}

Api& api() {
  if (!load()) throw std::runtime_error("HDF5: " + g_why);
  return g_api;
}

void chk(long long rc, const std::string& what) {
  if (rc < 0) throw std::runtime_error("HDF5: " + what + " failed");
}

// ids opened in a scope, closed in reverse order when the scope is left -- by return or by exception.  (A half-written .h5
// whose file id leaks stays open, and with HDF5 file locking LOCKED, in this process: the next output to that path fails too.)
struct Ids {
  typedef herr_t (*close_fn)(hid_t);
  std::vector<std::pair<close_fn, hid_t> > open;
  hid_t add(close_fn f, hid_t id) { if (id >= 0) open.push_back(std::make_pair(f, id)); return id; }
  // close now (checked by the caller); the destructor skips it
  herr_t close(hid_t id) {
    for (size_t i = open.size(); i-- > 0;)
      if (open[i].second == id) { const close_fn f = open[i].first; open.erase(open.begin() + (long)i); return f(id); }
    return -1;
  }
  ~Ids() { for (size_t i = open.size(); i-- > 0;) (void)open[i].first(open[i].second); }
};

struct Field { const char* name; int var; };
// datasets of a file in the order the reference writes them (HydroRunBase.cpp:3424-3510); component indices of the state
// arrays: ID, IP, IU, IV, IW, IA, IB, IC = 0..7
std::vector<Field> fields_of(const H5Box& b) {
  std::vector<Field> f = {{"/density", 0}, {"/energy", 1}, {"/momentum_x", 2}, {"/momentum_y", 3}};
  if (b.mhd || b.three_d) f.push_back({"/momentum_z", 4});
  if (b.mhd) { f.push_back({"/magnetic_field_x", 5}); f.push_back({"/magnetic_field_y", 6}); f.push_back({"/magnetic_field_z", 7}); }
  return f;
}

// memory space = the ghosted LOCAL array with planes [kmem0, kmem1) selected (all of x, y with ghosts, their interior
// without); file space = the whole box on disk (nz_file planes of the same x-y extent) with the same number of planes
// selected from plane kfile0 on.  2D: one "plane".
struct Spaces { hid_t mem, file; int rank; };
Spaces make_spaces(Api& a, Ids& ids, const H5Box& b, bool ghosts, hsize_t nz_file, hsize_t kmem0, hsize_t kmem1, hsize_t kfile0) {
  const hsize_t gw = (hsize_t)b.ghostWidth;
  const hsize_t full[3] = {(hsize_t)b.nz + 2 * gw, (hsize_t)b.ny + 2 * gw, (hsize_t)b.nx + 2 * gw};
  const hsize_t inner[3] = {(hsize_t)b.nz, (hsize_t)b.ny, (hsize_t)b.nx};
  const hsize_t* xy = ghosts ? full : inner;
  const int rank = b.three_d ? 3 : 2, o = b.three_d ? 0 : 1;   // 2D: (ny, nx)
  const hsize_t fdims[3] = {nz_file, xy[1], xy[2]};
  const hsize_t count[3] = {kmem1 - kmem0, xy[1], xy[2]};
  const hsize_t mstart[3] = {kmem0, ghosts ? 0 : gw, ghosts ? 0 : gw}, fstart[3] = {kfile0, 0, 0}, one[3] = {1, 1, 1};
  Spaces s;
  s.rank = rank;
  s.mem = ids.add(a.Sclose, a.Screate_simple(rank, full + o, 0));
  s.file = ids.add(a.Sclose, a.Screate_simple(rank, fdims + o, 0));
  chk(s.mem, "H5Screate_simple"); chk(s.file, "H5Screate_simple");
  chk(a.Sselect_hyperslab(s.mem, kSelectSet, mstart + o, one, count + o, one), "H5Sselect_hyperslab");
  chk(a.Sselect_hyperslab(s.file, kSelectSet, fstart + o, one, count + o, one), "H5Sselect_hyperslab");
  return s;
}
// the planes slab `r` of `n` contributes to / takes from a file of the whole box (see hdf5_io.h)
struct ZRange { hsize_t nz_file, kmem0, kmem1, kfile0; };
ZRange slab_range(const H5Box& b, int nz_global, int r, int n, bool ghosts, bool reading) {
  const hsize_t gw = (hsize_t)b.ghostWidth, nzl = (hsize_t)b.nz;
  ZRange z;
  if (!b.three_d) { z.nz_file = 1; z.kmem0 = 0; z.kmem1 = 1; z.kfile0 = 0; return z; }
  z.nz_file = (hsize_t)nz_global + (ghosts ? 2 * gw : 0);
  if (!ghosts) { z.kmem0 = gw; z.kmem1 = gw + nzl; z.kfile0 = (hsize_t)r * nzl; }
  else if (reading) { z.kmem0 = 0; z.kmem1 = nzl + 2 * gw; z.kfile0 = (hsize_t)r * nzl; }   // file plane = r * nzl + local plane
  else {
    z.kmem0 = (r == 0) ? 0 : gw;
    z.kmem1 = (r == n - 1) ? nzl + 2 * gw : nzl + gw;
    z.kfile0 = (hsize_t)r * nzl + z.kmem0;
  }
  return z;
}

template <class T>
void write_scalar_attr(Api& a, hid_t file, const char* name, hid_t type, const T& v) {
  Ids ids;
  const hid_t sp = ids.add(a.Sclose, a.Screate(kScalar));
  chk(sp, "H5Screate");
  const hid_t at = ids.add(a.Aclose, a.Acreate2(file, name, type, sp, kDefault, kDefault));
  chk(at, std::string("H5Acreate2 ") + name);
  chk(a.Awrite(at, type, &v), std::string("H5Awrite ") + name);
}

std::string current_date_utc() {
  char buf[64];
  const std::time_t t = std::time(0);
  std::strftime(buf, sizeof(buf), "%Y-%m-%d %H:%M:%S UTC", std::gmtime(&t));
  return buf;
}

}  // namespace

bool hdf5_available(std::string* why) {
  const bool ok = load();
  if (!ok && why) *why = g_why;
  return ok;
}

void hdf5_write_state(const std::string& path, const double* U, const H5Box& b, bool ghostIncluded, int nStep, double totalTime,
                      int compressionLevel) {
  hdf5_write_slab(path, U, b, b.nz, 0, 1, true, ghostIncluded, nStep, totalTime, compressionLevel);
}

void hdf5_write_slab(const std::string& path, const double* U, const H5Box& b, int nz_global, int slab_rank, int slab_count, bool create,
                     bool ghostIncluded, int nStep, double totalTime, int compressionLevel) {
  Api& a = api();
  if (compressionLevel < 0 || compressionLevel > 9) compressionLevel = 0;   // the reference warns and falls back to 0
  const size_t gw = (size_t)b.ghostWidth;
  const size_t ncell = (b.nx + 2 * gw) * (b.ny + 2 * gw) * (b.three_d ? b.nz + 2 * gw : 1);
  Ids ids;   // everything opened below is closed when this function is left, also by an exception
  const hid_t file = ids.add(a.Fclose, create ? a.Fcreate(path.c_str(), kAccTrunc, kDefault, kDefault) : a.Fopen(path.c_str(), kAccRdwr, kDefault));
  chk(file, (create ? "H5Fcreate " : "H5Fopen (read-write) ") + path);
  const ZRange z = slab_range(b, nz_global, slab_rank, slab_count, ghostIncluded, false);
  const Spaces sp = make_spaces(a, ids, b, ghostIncluded, z.nz_file, z.kmem0, z.kmem1, z.kfile0);
  hid_t dcpl = -1;
  if (create) {
    dcpl = ids.add(a.Pclose, a.Pcreate(a.cls_dataset_create));
    chk(dcpl, "H5Pcreate");
    const hsize_t chunk[3] = {(hsize_t)(b.three_d ? nz_global : 1), (hsize_t)b.ny, (hsize_t)b.nx};
    chk(a.Pset_chunk(dcpl, sp.rank, chunk + (b.three_d ? 0 : 1)), "H5Pset_chunk");
    chk(a.Pset_shuffle(dcpl), "H5Pset_shuffle");
    chk(a.Pset_deflate(dcpl, (unsigned)compressionLevel), "H5Pset_deflate");
  }
  // the extent of a dataset is that of the file space (the selection only says which part is written now)
  for (const Field& f : fields_of(b)) {
    const hid_t ds = ids.add(a.Dclose, create ? a.Dcreate2(file, f.name, a.native_double, sp.file, kDefault, dcpl, kDefault) : a.Dopen2(file, f.name, kDefault));
    chk(ds, std::string(create ? "H5Dcreate2 " : "H5Dopen2 ") + f.name);
    chk(a.Dwrite(ds, a.native_double, sp.mem, sp.file, kDefault, U + (size_t)f.var * ncell), std::string("H5Dwrite ") + f.name);
    chk(ids.close(ds), std::string("H5Dclose ") + f.name);
  }
  if (create) {
    write_scalar_attr(a, file, "time step", a.native_int, nStep);
    write_scalar_attr(a, file, "total time", a.native_double, totalTime);
    write_scalar_attr(a, file, "nx", a.native_int, b.nx);
    write_scalar_attr(a, file, "ny", a.native_int, b.ny);
    const int nzg = b.three_d ? nz_global : b.nz;
    write_scalar_attr(a, file, "nz", a.native_int, nzg);
    const int gi = ghostIncluded ? 1 : 0;
    write_scalar_attr(a, file, "ghost zone included", a.native_int, gi);
    // "creation date": one variable-length string
    const std::string date = current_date_utc();
    const char* ptr = date.c_str();
    const hid_t st = ids.add(a.Tclose, a.Tcopy(a.c_s1));
    chk(st, "H5Tcopy");
    chk(a.Tset_size(st, kVariable), "H5Tset_size");
    const hsize_t one = 1;
    const hid_t dsp = ids.add(a.Sclose, a.Screate_simple(1, &one, 0));
    chk(dsp, "H5Screate_simple");
    const hid_t at = ids.add(a.Aclose, a.Acreate2(file, "creation date", st, dsp, kDefault, kDefault));
    chk(at, "H5Acreate2 creation date");
    chk(a.Awrite(at, st, &ptr), "H5Awrite creation date");
    ids.close(at); ids.close(dsp); ids.close(st);
    ids.close(dcpl);
  }
  ids.close(sp.mem);
  ids.close(sp.file);
  chk(a.Fflush(file, kScopeLocal), "H5Fflush " + path);
  chk(ids.close(file), "H5Fclose " + path);
}

int hdf5_read_state(const std::string& path, double* U, const H5Box& b, double* totalTime, bool* ghostsInFile) {
  return hdf5_read_slab(path, U, b, b.nz, 0, 1, totalTime, ghostsInFile);
}

int hdf5_read_slab(const std::string& path, double* U, const H5Box& b, int nz_global, int slab_rank, int slab_count, double* totalTime,
                   bool* ghostsInFile) {
  Api& a = api();
  const size_t gw = (size_t)b.ghostWidth;
  const size_t ncell = (b.nx + 2 * gw) * (b.ny + 2 * gw) * (b.three_d ? b.nz + 2 * gw : 1);
  Ids ids;
  const hid_t file = ids.add(a.Fclose, a.Fopen(path.c_str(), kAccRdonly, kDefault));
  if (file < 0) throw std::runtime_error("restart: cannot open " + path + " as an HDF5 file");
  // with or without ghosts: decided by the extent of the first dataset (the reference trusts [output] ghostIncluded instead)
  bool ghosts = false;
  {
    const hid_t ds = ids.add(a.Dclose, a.Dopen2(file, "/density", kDefault));
    if (ds < 0) throw std::runtime_error("restart: " + path + " has no /density dataset");
    const hid_t fs = ids.add(a.Sclose, a.Dget_space(ds));
    chk(fs, "H5Dget_space /density");
    hsize_t dims[3] = {0, 0, 0};
    const int rank = a.Sget_simple_extent_ndims(fs);
    if (rank == (b.three_d ? 3 : 2)) chk(a.Sget_simple_extent_dims(fs, dims, 0), "H5Sget_simple_extent_dims /density");
    ids.close(fs);
    ids.close(ds);
    const hsize_t inner[3] = {(hsize_t)(b.three_d ? nz_global : 1), (hsize_t)b.ny, (hsize_t)b.nx};
    const int o = b.three_d ? 0 : 1, n = b.three_d ? 3 : 2;
    bool is_inner = rank == n, is_full = rank == n;
    for (int d = 0; d < n; ++d) {
      is_inner = is_inner && dims[d] == inner[o + d];
      is_full = is_full && dims[d] == inner[o + d] + 2 * gw;
    }
    if (!is_inner && !is_full) {
      std::ostringstream m;
      m << "restart: " << path << " holds another box than expected from [mesh] nx, ny, nz (/density is";
      for (int d = 0; d < rank && d < 3; ++d) m << " " << dims[d];
      m << ")";
      throw std::runtime_error(m.str());
    }
    ghosts = is_full;
  }
  const ZRange z = slab_range(b, nz_global, slab_rank, slab_count, ghosts, true);
  const Spaces sp = make_spaces(a, ids, b, ghosts, z.nz_file, z.kmem0, z.kmem1, z.kfile0);
  for (const Field& f : fields_of(b)) {
    const hid_t ds = ids.add(a.Dclose, a.Dopen2(file, f.name, kDefault));
    if (ds < 0) throw std::runtime_error("restart: " + path + " has no dataset " + f.name);
    const herr_t rc = a.Dread(ds, a.native_double, sp.mem, sp.file, kDefault, U + (size_t)f.var * ncell);
    ids.close(ds);
    if (rc < 0) throw std::runtime_error("restart: reading " + std::string(f.name) + " of " + path + " failed");
  }
  int step = 0;
  double t = 0.0;
  {
    // (files without these attributes resume at step 0, time 0; an attribute that is there must be readable)
    hid_t at = ids.add(a.Aclose, a.Aopen(file, "time step", kDefault));
    if (at >= 0) { chk(a.Aread(at, a.native_int, &step), "H5Aread time step"); ids.close(at); }
    at = ids.add(a.Aclose, a.Aopen(file, "total time", kDefault));
    if (at >= 0) { chk(a.Aread(at, a.native_double, &t), "H5Aread total time"); ids.close(at); }
  }
  if (totalTime) *totalTime = t;
  if (ghostsInFile) *ghostsInFile = ghosts;
  return step;
}

void xdmf_write_wrapper(const std::string& outputPrefix, const H5Box& b, bool ghostIncluded, int totalNumberOfSteps, int nOutput) {
  const int g2 = ghostIncluded ? 2 * b.ghostWidth : 0;
  const int nxg = b.nx + g2, nyg = b.ny + g2, nzg = b.nz + g2;
  std::ostringstream dims;
  if (b.three_d) dims << nzg << " " << nyg << " " << nxg; else dims << nyg << " " << nxg;
  const std::string D = dims.str();
  std::ofstream x((outputPrefix + ".xmf").c_str());
  if (!x) throw std::runtime_error("cannot write " + outputPrefix + ".xmf");
  x << "<?xml version=\"1.0\" ?>\n<!DOCTYPE Xdmf SYSTEM \"Xdmf.dtd\" []>\n"
    << "<Xdmf xmlns:xi=\"http://www.w3.org/2003/XInclude\" Version=\"2.2\">\n  <Domain>\n"
    << "    <Grid Name=\"TimeSeries\" GridType=\"Collection\" CollectionType=\"Temporal\">\n";
  const int nd = b.three_d ? 3 : 2;
  for (int n = 0; n <= totalNumberOfSteps && nOutput > 0; n += nOutput) {
    std::ostringstream num;
    num << std::setw(7) << std::setfill('0') << n;
    const std::string base = outputPrefix + "_" + num.str(), h5 = base + ".h5";
    x << "    <Grid Name=\"" << base << "\" GridType=\"Uniform\">\n    <Time Value=\"" << n << "\" />\n"
      << "      <Topology TopologyType=\"" << nd << "DCoRectMesh\" NumberOfElements=\"" << D << "\"/>\n"
      << "    <Geometry Type=\"" << (b.three_d ? "ORIGIN_DXDYDZ" : "ORIGIN_DXDY") << "\">\n";
    for (int pass = 0; pass < 2; ++pass) {
      x << "    <DataStructure\n       Name=\"" << (pass ? "Spacing" : "Origin") << "\"\n       DataType=\"Double\"\n       Dimensions=\"" << nd
        << "\"\n       Format=\"XML\">\n       ";
      for (int d = 0; d < nd; ++d) x << (d ? " " : "") << pass;
      x << "\n    </DataStructure>\n";
    }
    x << "    </Geometry>\n";
    for (const Field& f : fields_of(b)) {
      x << "      <Attribute Center=\"Node\" Name=\"" << (f.name + 1) << "\">\n        <DataStructure\n           DataType=\"Double\"\n"
        << "           Dimensions=\"" << D << "\"\n           Format=\"HDF\">\n           " << h5 << ":" << f.name
        << "\n        </DataStructure>\n      </Attribute>\n";
    }
    x << "   </Grid>\n";
  }
  x << "   </Grid>\n </Domain>\n</Xdmf>\n";
}

}  // namespace rgpu_host

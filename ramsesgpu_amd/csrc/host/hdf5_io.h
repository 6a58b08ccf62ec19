// hdf5_io.h -- the reference's HDF5 outputs and restart input for the run driver (host side, no device code):
//   outputHdf5               HydroRunBase.cpp:3308-3560   one file <prefix>_NNNNNNN.h5 per output, all fields in it
//   inputHdf5                HydroRunBase.cpp:4818-5155   restart (same resolution, or half of it for restart_upscale)
//   writeXdmfForHdf5Wrapper  HydroRunBase.cpp:3823-4073   <prefix>.xmf, the XDMF index of the .h5 files of a run
// File layout (so that files are interchangeable with the reference's, either way): datasets /density, /energy,
// /momentum_x, /momentum_y, /momentum_z (3D hydro and MHD), /magnetic_field_x, _y, _z (MHD); rank 2 (ny, nx) or 3
// (nz, ny, nx), doubles, x fastest, interior cells or -- [output] ghostIncluded -- the whole ghosted arrays; chunked
// (nz, ny, nx), shuffle + deflate at [output] outputHdf5CompressionLevel; root attributes "time step" (int), "total time"
// (double), "nx", "ny", "nz", "ghost zone included" (int), "creation date" (variable-length string).
//
// libhdf5 is bound at RUN time (dlopen: $RGPU_HDF5_LIB, libhdf5.so, libhdf5.so.103, /opt/conda/lib/libhdf5.so.103, ...):
// the product library does not link it, a machine without HDF5 loses only these two formats (the callers fall back to the
// raw dump and say so).  HDF5 >= 1.10 (64-bit hid_t).
#pragma once
#include <string>
#include <vector>

namespace rgpu_host {

struct H5Box {
  int nx, ny, nz;        // interior cells of the arrays to write / fill (nz = 1 in 2D)
  int ghostWidth, nbVar;
  bool three_d, mhd;
};

// true when libhdf5 could be loaded; otherwise *why says what was tried
bool hdf5_available(std::string* why);

// U: ghost-inclusive SoA arrays [nbVar][ksize][jsize][isize].  Throws std::runtime_error on failure.
void hdf5_write_state(const std::string& path, const double* U, const H5Box& b, bool ghostIncluded, int nStep, double totalTime,
                      int compressionLevel);

// Fills U (ghost-inclusive arrays of box b) from a file of the same box: the whole arrays if the file holds the ghosts, the
// interior otherwise (ghost cells of U untouched).  Returns the "time step" attribute; *totalTime, *ghostsInFile set.
int hdf5_read_state(const std::string& path, double* U, const H5Box& b, double* totalTime, bool* ghostsInFile);

// z-slab runs (one process per slab, 3D): ONE file for the whole box, identical to the single-domain file.  b describes the
// LOCAL slab (b.nz = its planes); the file holds nz_global planes.  Serial HDF5: the ranks take turns -- rank 0 first, with
// create = true (creates the file, the datasets at the extents of the whole box, the attributes), then every other rank opens
// it read-write and adds the planes it owns (its interior planes; with ghostIncluded the first / last rank also write the low /
// high z ghost planes).  The caller provides the turn taking (a barrier between ranks).
void hdf5_write_slab(const std::string& path, const double* U, const H5Box& b, int nz_global, int slab_rank, int slab_count, bool create,
                     bool ghostIncluded, int nStep, double totalTime, int compressionLevel);
// every rank reads its planes (with its z ghost planes when the file holds ghosts) from the file of the whole box
int hdf5_read_slab(const std::string& path, double* U, const H5Box& b, int nz_global, int slab_rank, int slab_count, double* totalTime,
                   bool* ghostsInFile);

// <prefix>.xmf in the CURRENT directory (as the reference does), entries for steps 0, nOutput, ... <= totalNumberOfSteps
void xdmf_write_wrapper(const std::string& outputPrefix, const H5Box& b, bool ghostIncluded, int totalNumberOfSteps, int nOutput);

}  // namespace rgpu_host

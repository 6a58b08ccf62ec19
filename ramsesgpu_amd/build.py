"""Build the product library ramsesgpu_amd/librgpu.so (HIP, gfx950 only) and the euler_hip front end, in-tree.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off ...

-ffp-contract=off is part of the parity contract: the reference CPU path is built without FMA contraction and the
kernels reproduce its operand order, so results are bit-identical ("Exactness").
hipcc cross-compiles without a GPU, so this also runs in the build container.

Second, opt-in variant of the same sources: librgpu_fast.so (+ librgpu_comm_fast.so), "contracted" arithmetic --
-ffp-contract=fast -DRG_ARITH_FAST=2: mul+add pairs fuse into FMAs, the shared-reciprocal division and the square root
drop their last correction step (results within ~1 ulp instead of correctly rounded).  Same operand order, same
algorithms; results agree with the reference to round-off (worst relative L2 over all golden fixtures 2e-14, stated
tolerance 1e-12), not bit for bit.  Same C ABI, same symbol names: a host program links -lrgpu OR -lrgpu_fast.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

HOST_SRC = ["ini_config.cpp", "host_params.cpp", "init_conditions.cpp", "host_capi.cpp", "run_driver.cpp", "hdf5_io.cpp"]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", os.path.join(CSRC, "hip"), "-I", CSRC]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


FAST_FLAGS = ["-DRG_ARITH_FAST=2", "-ffp-contract=fast"]


def resources_path(out_name="librgpu.so"):
    """the compiler's kernel-resource-usage remarks of the last device compile of `out_name`"""
    return os.path.join(HERE, "..", "build", "obj_" + out_name.replace(".", "_"), "kernel_resources.txt")


def kernel_source_hash():
    """sha256 (16 hex digits) of the device sources -- csrc/*.h, csrc/hip/*.h, csrc/api/*.h, csrc/rgpu_api.cpp with // comments and whitespace runs
    removed -- so that numbers MEASURED on one state of the kernels (profiles/pmc_traffic.json: instruction counts, HBM bytes) are
    not quoted for another: scripts/prof_round.sh records it with the counters, bench.py compares before it uses them"""
    import hashlib
    import re
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + \
        [os.path.join(CSRC, "hip", f) for f in sorted(os.listdir(os.path.join(CSRC, "hip"))) if f.endswith(".h") and f not in ("rg_transport.h", "halo_pack.h")] + \
        [os.path.join(CSRC, "api", f) for f in sorted(os.listdir(os.path.join(CSRC, "api"))) if f.endswith(".h")] + [os.path.join(CSRC, "rgpu_api.cpp")]
    # (rg_transport.h / halo_pack.h are the RCCL transport of librgpu_comm.so: no kernel of librgpu.so comes from them)
    h = hashlib.sha256()
    for f in files:
        text = re.sub(r"//[^\n]*", "", open(f).read())
        h.update(os.path.basename(f).encode())
        h.update(re.sub(r"\s+", " ", text).encode())
    return h.hexdigest()[:16]


def kernel_resources(out_name="librgpu.so"):
    """{demangled kernel name: {"vgprs", "agprs", "sgprs", "sgpr_spill", "vgpr_spill", "scratch", "lds", "occupancy"}} parsed from
    resources_path(out_name); {} when the library was built without it (an older build.py)"""
    import re
    path = resources_path(out_name)
    if not os.path.exists(path):
        return {}
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
            "ScratchSize [bytes/lane]": "scratch", "LDS Size [bytes/block]": "lds", "Occupancy [waves/SIMD]": "occupancy"}
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    mangled = list(out)
    if mangled:
        try:
            dem = subprocess.run(["c++filt"], input="\n".join(mangled), stdout=subprocess.PIPE, universal_newlines=True).stdout.split("\n")
            out = {d: out[m] for m, d in zip(mangled, dem)}
        except OSError:
            pass
    return out


def build(verbose=True, force=False, extra_flags=(), out_name="librgpu.so"):
    """out_name "librgpu.so": the product (+ librgpu_comm.so, euler_hip); "librgpu_fast.so": the contracted-arithmetic
    variant (FAST_FLAGS are added, + librgpu_comm_fast.so); any other name: an experiment build of the library alone"""
    out = os.path.join(HERE, out_name)
    fast = out_name == "librgpu_fast.so"
    if fast:
        extra_flags = list(extra_flags) + FAST_FLAGS
    exe = os.path.join(HERE, "euler_hip")
    srcs = [os.path.join(CSRC, "rgpu_api.cpp")] + [os.path.join(CSRC, "host", s) for s in HOST_SRC]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(CSRC, "api", f) for f in os.listdir(os.path.join(CSRC, "api")) if f.endswith(".h")] + \
        [os.path.join(CSRC, "hip", f) for f in os.listdir(os.path.join(CSRC, "hip")) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "rgpu.h")] + \
        [os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host")) if f.endswith(".h")]
    if force or _newer(out, deps):
        # one object per translation unit, then one link: mixing "-x hip" and "-x c++" inputs in a single hipcc
        # invocation produced a library whose kernels crashed at launch (ROCm 7.2), so keep them apart
        objdir = os.path.join(HERE, "..", "build", "obj_" + out_name.replace(".", "_"))
        os.makedirs(objdir, exist_ok=True)
        objs = []
        for i, src in enumerate(srcs):
            obj = os.path.join(objdir, os.path.basename(src) + ".o")
            if i == 0:
                # the device compile also reports what every kernel needs (registers, spills, scratch, LDS, occupancy): kept next
                # to the objects and checked by tests/test_kernel_resources.py -- the speed of the sweeps hangs on these numbers
                cmd = [HIPCC, "--offload-arch=" + ARCH] + COMMON + list(extra_flags) + ["-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c", src, "-o", obj]
            else:
                cmd = [HIPCC] + COMMON + ["-x", "c++", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            if i == 0:
                r = subprocess.run(cmd, stderr=subprocess.PIPE, universal_newlines=True)
                with open(resources_path(out_name), "w") as f:
                    f.write(r.stderr)
                if r.returncode != 0:
                    sys.stderr.write(r.stderr[-20000:])
                    raise subprocess.CalledProcessError(r.returncode, cmd)
            else:
                subprocess.check_call(cmd)
            objs.append(obj)
        # -Bsymbolic-functions: calls between the library's own exported functions bind inside the library (librgpu.so and
        # librgpu_fast.so export the same names and may share a process)
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,-Bsymbolic-functions"] + objs + ["-ldl", "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # z-slab driver: librgpu_comm.so = csrc/comm/rgpu_comm.cpp + the RCCL transport (csrc/hip/rg_transport.h)
    comm_src = os.path.join(CSRC, "comm", "rgpu_comm.cpp")
    comm_out = os.path.join(HERE, "librgpu_comm_fast.so" if fast else "librgpu_comm.so")
    comm_deps = [comm_src, os.path.join(CSRC, "hip", "rg_transport.h"), os.path.join(CSRC, "hip", "halo_pack.h"), os.path.join(CSRC, "comm", "halo_ops.h"), os.path.join(CSRC, "comm", "pack_plan.h"),
                 os.path.join(HERE, "..", "include", "rgpu_comm.h"),
                 os.path.join(HERE, "..", "include", "rgpu.h"), out]
    if (out_name == "librgpu.so" or fast) and (force or _newer(comm_out, comm_deps)):
        # (-x hip: halo_pack.h holds the pack / unpack kernels of the packed exchange)
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic-functions", "-I", os.path.join(CSRC, "hip"), "-x", "hip", comm_src, "-x", "none", "-L", HERE, "-lrgpu_fast" if fast else "-lrgpu",
               "-L", "/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN", "-o", comm_out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    main_src = os.path.join(CSRC, "host", "euler_hip_main.cpp")
    if out_name == "librgpu.so" and (force or _newer(exe, [main_src, out, comm_out])):
        cmd = [HIPCC, "-O2", "-std=c++17", main_src, "-L", HERE, "-lrgpu_comm", "-lrgpu", "-Wl,-rpath,$ORIGIN", "-o", exe]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build_measure(verbose=True, fast=True):
    """MEASUREMENT build of the slab driver, NOT the product: librgpu_comm_measure.so = csrc/comm/rgpu_comm.cpp with -DRG_MEASURE and
    scripts/measure/link_hold.h (the emulated xGMI link time of the one-GPU slab probe, scripts/slab_probe.py), linked against the
    product library librgpu_fast.so / librgpu.so"""
    build(verbose=verbose, out_name="librgpu_fast.so" if fast else "librgpu.so")
    out = os.path.join(HERE, "librgpu_comm_measure.so")
    comm_src = os.path.join(CSRC, "comm", "rgpu_comm.cpp")
    cmd = [HIPCC, "--offload-arch=" + ARCH, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic-functions", "-DRG_MEASURE", "-I", os.path.join(HERE, "..", "scripts", "measure"),
           "-I", os.path.join(CSRC, "hip"), "-x", "hip", comm_src, "-x", "none", "-L", HERE, "-lrgpu_fast" if fast else "-lrgpu",
           "-L", "/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN", "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_all(verbose=True, force=False):
    """the product library and its contracted-arithmetic variant"""
    out = build(verbose=verbose, force=force)
    build(verbose=verbose, force=force, out_name="librgpu_fast.so")
    return out


if __name__ == "__main__":
    if "--measure" in sys.argv:
        build_measure()
    else:
        build_all(force="--force" in sys.argv)

"""The packed halo exchange of the RCCL transport (csrc/hip/rg_transport.h, round 4) for rank counts a one-GPU box cannot run: the
operation list of every slab (csrc/comm/halo_ops.h) and the staging plan (csrc/comm/pack_plan.h) are plain C++; tests/cpp/pack_plan_check.cpp
carries the exchange out on the host -- pack, ONE message per ordered pair of neighbours, unpack -- and checks every ghost plane."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("packplan") / "pack_plan_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "ramsesgpu_amd", "csrc", "comm"),
                           os.path.join(ROOT, "tests", "cpp", "pack_plan_check.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("nranks,periodic", [(1, 1), (2, 1), (3, 1), (5, 1), (8, 1), (1, 0), (2, 0), (3, 0), (8, 0)])
@pytest.mark.parametrize("nvar,nz", [(8, 7), (5, 3)])
def test_packed_exchange_on_the_host(checker, nranks, periodic, nvar, nz):
    out = subprocess.run([checker, str(nranks), str(periodic), str(nvar), str(nz)], stdout=subprocess.PIPE, universal_newlines=True)
    assert out.returncode == 0, out.stdout

#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING THE REFERENCE ITSELF (oracle/_ref/euler_cpu).

Test infrastructure only.  Runs in the build container (where /root/reference exists and
`make -C oracle -f Makefile.ref` has produced oracle/_ref/euler_cpu).  For every case in CASES it

  1. writes a temporary .ini = configs/<base>.ini with the case's overrides applied,
  2. runs the reference `euler_cpu --param tmp.ini` (g++ -O2, no OpenMP, no FMA: see Makefile.ref),
  3. parses the hand-written VTI outputs (HydroRunBase.cpp:2681-2995: appended raw, uint32 byte count +
     nx*ny*nz little-endian Float64 per variable, INTERIOR cells only),
  4. stores the requested steps + the per-step dt log + the final time in tests/golden/<case>.npz.

The fixtures are DATA (inputs = configs + overrides listed in tests/golden/cases.json, outputs = arrays).
No reference source text is stored.

usage: python oracle/gen_golden.py [case ...]
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_BIN = os.path.join(HERE, "_ref", "euler_cpu")
GOLDEN = os.path.join(ROOT, "tests", "golden")

# name -> (base config, overrides, steps to keep)
# every run uses nlog=1 so that the dt of every step is printed with 12 decimals (MHDRunGodunov.cpp:3925-3931)
CASES = {
    # --- 2D MHD (implementation 1): Orszag-Tang (periodic) ------------------------------------------------
    "ot2d_32": ("orszag-tang", "mesh.nx=32;mesh.ny=32;run.nstepmax=50;run.noutput=10", [0, 10, 50]),
    "ot2d_32_s1": ("orszag-tang", "mesh.nx=32;mesh.ny=32;run.nstepmax=1;run.noutput=10", [1]),
    "ot2d_64": ("orszag-tang", "mesh.nx=64;mesh.ny=64;run.nstepmax=20;run.noutput=100", [20]),
    "ot2d_48x24": ("orszag-tang", "mesh.nx=48;mesh.ny=24;run.nstepmax=7;run.noutput=100", [7]),
    # --- 2D MHD Brio-Wu (Neumann) ---------------------------------------------------------------------------
    "briowu_x_64": ("mhd_BrioWu", "mesh.nx=64;mesh.ny=64;BrioWu.direction=0;run.nstepmax=20;run.noutput=100", [0, 20]),
    "briowu_y_64": ("mhd_BrioWu", "mesh.nx=64;mesh.ny=64;BrioWu.direction=1;run.nstepmax=20;run.noutput=100", [0, 20]),
    # --- 3D hydro implode (Dirichlet), the three hydro Riemann solvers ---------------------------------------
    "implode3d_16_approx": ("implode3d", "mesh.nx=16;mesh.ny=16;mesh.nz=16;run.nstepmax=10;run.noutput=100", [0, 10]),
    "implode3d_16_hllc": ("implode3d", "mesh.nx=16;mesh.ny=16;mesh.nz=16;hydro.riemannSolver=hllc;run.nstepmax=10;run.noutput=100", [10]),
    "implode3d_16_hll": ("implode3d", "mesh.nx=16;mesh.ny=16;mesh.nz=16;hydro.riemannSolver=hll;run.nstepmax=10;run.noutput=100", [10]),
    "implode3d_24x16x12_hllc_mc": ("implode3d", "mesh.nx=24;mesh.ny=16;mesh.nz=12;hydro.riemannSolver=hllc;hydro.slope_type=2.0;run.nstepmax=8;run.noutput=100", [8]),
    # --- 2D hydro: implode 2D and the jet (inflow BC) ----------------------------------------------------------
    "implode2d_32_hllc": ("implode3d", "mesh.nx=32;mesh.ny=32;mesh.nz=1;hydro.riemannSolver=hllc;run.nstepmax=10;run.noutput=100", [0, 10]),
    "implode2d_32_approx": ("implode3d", "mesh.nx=32;mesh.ny=32;mesh.nz=1;run.nstepmax=10;run.noutput=100", [10]),
    "jet2d_20x80": ("jet2d_cpu", "mesh.nx=20;mesh.ny=80;jet.ijet=4;jet.offsetJet=3;run.nstepmax=20;run.noutput=100", [0, 20]),
    # --- 3D MHD plain path (implementation 3/4) ----------------------------------------------------------------
    "ot3d_16": ("orszag-tang3d", "mesh.nx=16;mesh.ny=16;mesh.nz=16;run.nstepmax=5;run.noutput=100", [0, 5]),
    "ot3d_16_s1": ("orszag-tang3d", "mesh.nx=16;mesh.ny=16;mesh.nz=16;run.nstepmax=1;run.noutput=100", [1]),
    "ot3d_20x12x8_kt": ("orszag-tang3d", "mesh.nx=20;mesh.ny=12;mesh.nz=8;OrszagTang.kt=1.0;run.nstepmax=4;run.noutput=100", [4]),
    # --- 3D MHD rotating + shearing box (MRI), the shipped size -------------------------------------------------
    "mri_16x32x16": ("mhd_mri_3d", "run.nstepmax=20;run.noutput=5", [0, 5, 20]),
    "mri_16x32x16_s1": ("mhd_mri_3d", "run.nstepmax=1;run.noutput=100", [1]),
    "mri_8x16x8_long": ("mhd_mri_3d", "mesh.nx=8;mesh.ny=16;mesh.nz=8;run.nstepmax=60;run.noutput=1000", [60]),
    # --- the other 2D magnetic Riemann solvers (riemann_mhd.h:417-609) and slope_type 3 -----------
    "ot2d_32_hlla": ("orszag-tang", "mesh.nx=32;mesh.ny=32;MHD.magRiemannSolver=hlla;run.nstepmax=20;run.noutput=100", [20]),
    "ot2d_32_hllf": ("orszag-tang", "mesh.nx=32;mesh.ny=32;MHD.magRiemannSolver=hllf;run.nstepmax=20;run.noutput=100", [20]),
    "ot2d_32_llf": ("orszag-tang", "mesh.nx=32;mesh.ny=32;MHD.magRiemannSolver=llf;run.nstepmax=20;run.noutput=100", [20]),
    "ot3d_12_hlla": ("orszag-tang3d", "mesh.nx=12;mesh.ny=12;mesh.nz=12;MHD.magRiemannSolver=hlla;run.nstepmax=4;run.noutput=100", [4]),
    "ot3d_12_llf": ("orszag-tang3d", "mesh.nx=12;mesh.ny=12;mesh.nz=12;MHD.magRiemannSolver=llf;run.nstepmax=4;run.noutput=100", [4]),
    "mri_8x16x8_hllf": ("mhd_mri_3d", "mesh.nx=8;mesh.ny=16;mesh.nz=8;MHD.magRiemannSolver=hllf;run.nstepmax=10;run.noutput=1000", [10]),
    "ot2d_32_slope3": ("orszag-tang", "mesh.nx=32;mesh.ny=32;hydro.slope_type=3.0;run.nstepmax=20;run.noutput=100", [20]),
    "briowu_x_64_slope3": ("mhd_BrioWu", "mesh.nx=64;mesh.ny=64;BrioWu.direction=0;hydro.slope_type=3.0;run.nstepmax=20;run.noutput=100", [20]),
    "ot3d_12_slope3": ("orszag-tang3d", "mesh.nx=12;mesh.ny=12;mesh.nz=12;hydro.slope_type=3.0;run.nstepmax=4;run.noutput=100", [4]),
    # --- further shipped problems (initial conditions of the host side + the same step) -------------------------------
    # --- 2D branch of the rotating-frame step (godunov_unsplit_rotating_cpu, MHDRunGodunov.cpp:2089-2434) ---------------
    "inertialwave2d_32": ("mhd_inertialWave_2d", "run.nstepmax=30;run.noutput=1000;output.outputVtk=yes;output.outputHdf5=no;output.outputVtkAscii=no", [0, 30]),
    "ot2d_32_rot": ("orszag-tang", "mesh.nx=32;mesh.ny=32;MHD.omega0=0.5;run.nstepmax=20;run.noutput=100", [20]),
    "ot2d_24x40_rot_iso_hll": ("orszag-tang", "mesh.nx=24;mesh.ny=40;MHD.omega0=0.3;hydro.cIso=0.8;hydro.riemannSolver=hll;MHD.magRiemannSolver=hllf;run.nstepmax=12;run.noutput=100", [12]),
    "briowu_x_48_rot_open": ("mhd_BrioWu", "mesh.nx=48;mesh.ny=16;BrioWu.direction=0;MHD.omega0=0.2;hydro.nu=0.002;MHD.eta=0.004;run.nstepmax=15;run.noutput=100", [15]),
    "ot2d_20x16_rot_slope3": ("orszag-tang", "mesh.nx=20;mesh.ny=16;MHD.omega0=0.4;hydro.slope_type=3.0;MHD.magRiemannSolver=hlla;run.nstepmax=10;run.noutput=100", [10]),
    # --- per-cell static gravity field (h_gravity filled by the problem): Keplerian disk, 2D analytic / 3D differenced ----
    "kepler2d_32": ("Keplerian_disk2d", "mesh.nx=32;mesh.ny=32;run.nstepmax=15;run.noutput=1000", [0, 15]),
    "kepler2d_24x40_hll": ("Keplerian_disk2d", "mesh.nx=24;mesh.ny=40;hydro.riemannSolver=hll;hydro.unsplitVersion=2;run.nstepmax=10;run.noutput=1000", [10]),
    "kepler3d_16x16x6": ("Keplerian_disk2d", "mesh.nx=16;mesh.ny=16;mesh.nz=6;run.nstepmax=8;run.noutput=1000", [0, 8]),
    # --- vertically stratified MRI box: gravity field g_z(z), BC_Z_STRATIFIED, stratified initial condition ------------------
    # (slope_type overridden: with the shipped 3.0 the reference's rotating step uses slopes it never sets)
    "mri_strat_8x16x32": ("mhd_mri_3d_stratified", "mesh.nx=8;mesh.ny=16;mesh.nz=32;hydro.slope_type=2.0;output.ghostIncluded=no;output.outputVtkAscii=no;run.nstepmax=12;run.noutput=6", [0, 6, 12]),
    "mri_strat_6x8x24_floor": ("mhd_mri_3d_stratified", "mesh.nx=6;mesh.ny=8;mesh.nz=24;hydro.slope_type=1.0;MRI.floor=yes;MRI.smoothGravity=no;MRI.amp=0.3;MRI.zFloor=2.5;hydro.nu=1e-5;MHD.eta=2e-5;output.ghostIncluded=no;output.outputVtkAscii=no;run.nstepmax=8;run.noutput=1000", [8]),
    # --- driven turbulence: static driving field + normalisation sums (sequential in the reference: the device agrees to round-off) ---
    "turb_hydro_16": ("turbulence_hydro", "mesh.nx=16;mesh.ny=16;mesh.nz=16;output.outputVtk=yes;output.outputHdf5=no;output.ghostIncluded=no;run.nstepmax=10;run.noutput=1000", [0, 10]),
    "turb_hydro_12x12x18_hllc": ("turbulence_hydro", "mesh.nx=12;mesh.ny=12;mesh.nz=18;hydro.riemannSolver=hllc;hydro.unsplitVersion=2;turbulence.edot=-1.0;output.outputVtk=yes;output.outputHdf5=no;output.ghostIncluded=no;run.nstepmax=6;run.noutput=1000", [6]),
    # --- Ornstein-Uhlenbeck forcing: the host process is reproduced exactly; the device evaluates cos() itself (round-off agreement) ---
    "turb_ou_hydro_16": ("turbulence_hydro_ou", "mesh.nx=16;mesh.ny=16;mesh.nz=16;turbulence-Ornstein-Uhlenbeck.initialDensityPerturbationAmplitude=0.1;output.outputVtk=yes;output.outputHdf5=no;output.ghostIncluded=no;run.nstepmax=10;run.noutput=1000", [0, 10]),
    "turb_ou_hydro_12x16x20_hllc_ksi": ("turbulence_hydro_ou", "mesh.nx=12;mesh.ny=16;mesh.nz=20;hydro.riemannSolver=hllc;hydro.cIso=0;turbulence-Ornstein-Uhlenbeck.ksi=0.3;turbulence-Ornstein-Uhlenbeck.init_random=77;output.outputVtk=yes;output.outputHdf5=no;output.ghostIncluded=no;run.nstepmax=6;run.noutput=1000", [6]),
    "turb_ou_mhd_12": ("turbulence_mhd_ou", "mesh.nx=12;mesh.ny=12;mesh.nz=12;history.enabled=no;output.outputVtk=yes;output.outputHdf5=no;output.ghostIncluded=no;run.nstepmax=6;run.noutput=1000", [0, 6]),
    "turb_ou_mhd_12_history": ("turbulence_mhd_ou", "mesh.nx=12;mesh.ny=12;mesh.nz=12;history.dtHist=0.02;output.outputVtk=yes;output.outputHdf5=no;output.ghostIncluded=no;run.nstepmax=8;run.noutput=1000", [8]),
    "turb_mhd_12": ("turbulence_mhd", "mesh.nx=12;mesh.ny=12;mesh.nz=12;output.outputVtk=yes;output.outputHdf5=no;output.ghostIncluded=no;run.nstepmax=6;run.noutput=1000", [0, 6]),
    "rotor_32_ic": ("mhd_rotor", "mesh.nx=32;mesh.ny=32;run.nstepmax=0;run.noutput=100", [0]),   # IC only: with implementationVersion=1 the reference itself turns this problem into NaN within a few steps
    "fieldloop2d_32x20": ("mhd_fieldloop2d", "mesh.nx=32;mesh.ny=20;run.nstepmax=10;run.noutput=100", [0, 10]),
    "fieldloop3d_16x8x8": ("mhd_fieldloop3d", "mesh.nx=16;mesh.ny=8;mesh.nz=8;run.nstepmax=5;run.noutput=100", [0, 5]),
    "currentsheet2d_24": ("mhd_currentSheet_2d", "mesh.nx=24;mesh.ny=24;run.nstepmax=10;run.noutput=100", [0, 10]),
    "currentsheet3d_12x12x8": ("mhd_currentSheet_3d", "mesh.nx=12;mesh.ny=12;mesh.nz=8;run.nstepmax=5;run.noutput=100", [0, 5]),
    "blast2d_24x36": ("blast2d", "mesh.nx=24;mesh.ny=36;run.nstepmax=10;run.noutput=100", [0, 10]),
    "blast3d_12": ("blast2d", "mesh.nx=12;mesh.ny=18;mesh.nz=12;run.nstepmax=6;run.noutput=100", [0, 6]),
    "mhdjet2d_24": ("mhd_jet2d", "mesh.nx=24;mesh.ny=24;jet.ijet=4;jet.offsetJet=6;MHD.implementationVersion=1;run.nstepmax=12;run.noutput=100", [0, 12]),
    "mhdjet2d_24_bfield": ("mhd_jet2d", "mesh.nx=24;mesh.ny=24;jet.ijet=4;jet.offsetJet=6;jet.BStatic_y=0.5;jet.BStatic_x=0.1;MHD.implementationVersion=1;run.nstepmax=12;run.noutput=100", [12]),
    "mhdjet3d_10x10x20": ("mhd_jet3d", "mesh.nx=10;mesh.ny=10;mesh.nz=20;jet.ijet=3;jet.offsetJet=3;jet.BStatic_z=0.3;output.outputVtkAscii=no;run.nstepmax=8;run.noutput=100", [0, 8]),
    "kh2d_rand_24": ("kelvin_helmholtz_cpu_2d", "mesh.nx=24;mesh.ny=24;run.nstepmax=10;run.noutput=100", [0, 10]),
    "kh2d_robertson_32x24": ("kelvin_helmholtz_gpu_2d", "mesh.nx=32;mesh.ny=24;hydro.unsplitVersion=1;run.nstepmax=10;run.noutput=100", [0, 10]),
    "kh2d_sine_24": ("kelvin_helmholtz_cpu_2d", "mesh.nx=24;mesh.ny=24;kelvin-helmholtz.perturbation_rand=no;kelvin-helmholtz.perturbation_sine=yes;kelvin-helmholtz.outer_size=0.3;run.nstepmax=6;run.noutput=100", [0, 6]),
    "kh2d_athena_24": ("kelvin_helmholtz_cpu_2d", "mesh.nx=24;mesh.ny=24;kelvin-helmholtz.perturbation_rand=no;kelvin-helmholtz.perturbation_sine_athena=yes;run.nstepmax=6;run.noutput=100", [0, 6]),
    "kh3d_rand_16x4x16": ("kelvin_helmholtz_gpu_3d", "mesh.nx=16;mesh.ny=4;mesh.nz=16;run.nstepmax=6;run.noutput=100", [0, 6]),
    "kh3d_robertson_16x4x12": ("kelvin_helmholtz_gpu_3d", "mesh.nx=16;mesh.ny=4;mesh.nz=12;kelvin-helmholtz.perturbation_rand=no;kelvin-helmholtz.perturbation_sine_robertson=yes;run.nstepmax=5;run.noutput=100", [0, 5]),
    "mhdkh2d_24": ("mhd_kelvin_helmholtz_2d", "mesh.nx=24;mesh.ny=24;run.nstepmax=10;run.noutput=100", [0, 10]),
    # --- static gravity (predictor on the traced states + momentum source) ---------------------------
    "rt2d_16x48": ("rayleigh_taylor_gpu_2d", "mesh.nx=16;mesh.ny=48;run.nstepmax=12;run.noutput=100", [0, 12]),
    "rt2d_16x48_hllc_rand": ("rayleigh_taylor_gpu_2d", "mesh.nx=16;mesh.ny=48;hydro.riemannSolver=hllc;rayleigh-taylor.randomEnabled=yes;gravity.static_field_x=0.03;run.nstepmax=10;run.noutput=100", [0, 10]),
    "rt3d_8x8x32": ("rayleigh_taylor_gpu_3d", "mesh.nx=8;mesh.ny=8;mesh.nz=32;run.nstepmax=8;run.noutput=100", [0, 8]),
    "rt3d_mhd_8x8x24": ("rayleigh_taylor_gpu_3d_mhd", "mesh.nx=8;mesh.ny=8;mesh.nz=24;rayleigh-taylor.bx=0.05;run.nstepmax=6;run.noutput=100", [0, 6]),
    "rt2d_mhd_12x36": ("rayleigh_taylor_cpu_2d_mhd", "mesh.nx=12;mesh.ny=36;MHD.implementationVersion=1;run.nstepmax=8;run.noutput=100", [0, 8]),
    "implode3d_12_gravity": ("implode3d", "mesh.nx=12;mesh.ny=12;mesh.nz=12;hydro.riemannSolver=hllc;gravity.static=yes;gravity.static_field_x=0.2;gravity.static_field_y=-0.1;gravity.static_field_z=0.4;run.nstepmax=6;run.noutput=100", [6]),
    "ot3d_12_gravity": ("orszag-tang3d", "mesh.nx=12;mesh.ny=12;mesh.nz=12;gravity.static=yes;gravity.static_field_x=0.1;gravity.static_field_y=0.2;gravity.static_field_z=-0.3;run.nstepmax=4;run.noutput=100", [4]),
    "gresho2d_32": ("Gresho_vortex2d", "mesh.nx=32;mesh.ny=32;hydro.unsplitVersion=1;run.nstepmax=10;run.noutput=100", [0, 10]),
    "gresho3d_16x16x6": ("Gresho_vortex2d", "mesh.nx=16;mesh.ny=16;mesh.nz=6;hydro.unsplitVersion=1;Gresho_vortex.v_bulk_z=0.25;run.nstepmax=5;run.noutput=100", [0, 5]),
    "bubble2d_24": ("falling_bubble_gpu_2d", "mesh.nx=24;mesh.ny=24;run.nstepmax=10;run.noutput=100", [0, 10]),
    "shearwave_16x8x4": ("mhd_shearWave_3d", "mesh.nx=16;mesh.ny=8;mesh.nz=4;output.outputVtkAscii=no;run.nstepmax=8;run.noutput=100", [0, 8]),
    # --- hydro unsplitVersion=2 (direction-wise sweeps: same fluxes, other accumulation order) ---------------------
    "gresho2d_32_v2": ("Gresho_vortex2d", "mesh.nx=32;mesh.ny=32;run.nstepmax=10;run.noutput=100", [10]),
    "kh2d_robertson_32x24_v2": ("kelvin_helmholtz_gpu_2d", "mesh.nx=32;mesh.ny=24;run.nstepmax=10;run.noutput=100", [10]),
    "implode3d_12_v2": ("implode3d", "mesh.nx=12;mesh.ny=12;mesh.nz=12;hydro.unsplitVersion=2;hydro.riemannSolver=hllc;run.nstepmax=6;run.noutput=100", [6]),
    "rt2d_16x48_v2": ("rayleigh_taylor_gpu_2d", "mesh.nx=16;mesh.ny=48;hydro.unsplitVersion=2;run.nstepmax=8;run.noutput=100", [8]),
    # --- 2D MHD implementationVersion=0 (same numbers as version 1, plus the gravity terms) ----------------------------
    "rt2d_mhd_12x36_v0": ("rayleigh_taylor_cpu_2d_mhd", "mesh.nx=12;mesh.ny=36;run.nstepmax=8;run.noutput=100", [8]),
    "briowu_x_32_v0": ("mhd_BrioWu", "mesh.nx=32;mesh.ny=16;BrioWu.direction=0;MHD.implementationVersion=0;run.nstepmax=12;run.noutput=100", [12]),
    "ot2d_24_v0": ("orszag-tang", "mesh.nx=24;mesh.ny=24;MHD.implementationVersion=0;run.nstepmax=12;run.noutput=100", [12]),
    "implode3d_12_rand": ("implode3d", "mesh.nx=12;mesh.ny=12;mesh.nz=12;implode.amplitude=0.02;implode.seed=7;hydro.riemannSolver=hllc;run.nstepmax=5;run.noutput=100", [0, 5]),
    "implode2d_16_rand": ("implode3d", "mesh.nx=16;mesh.ny=16;mesh.nz=1;implode.amplitude=0.05;run.nstepmax=5;run.noutput=100", [0, 5]),
    "riemann2d_c3_24": ("riemann2d", "mesh.nx=24;mesh.ny=24;run.nstepmax=8;run.noutput=100", [0, 8]),
    "riemann2d_c12_24": ("riemann2d", "mesh.nx=24;mesh.ny=24;hydro.riemann_config_number=11;riemann2d.x=0.5;riemann2d.y=0.4;run.nstepmax=8;run.noutput=100", [0, 8]),
    "riemann2d_c19_16": ("riemann2d", "mesh.nx=16;mesh.ny=16;hydro.riemann_config_number=25;run.nstepmax=4;run.noutput=100", [0, 4]),
    # --- history diagnostics: the run also writes <prefix>_history.txt (one row per step with dtHist=0) ------------------
    "mri_8x16x8_history": ("mhd_mri_3d", "mesh.nx=8;mesh.ny=16;mesh.nz=8;MRI.amp=0.2;history.enabled=yes;history.dtHist=0.0;run.nstepmax=10;run.noutput=1000", [10]),
    "inertialwave2d_16_history": ("mhd_inertialWave_2d", "mesh.nx=16;mesh.ny=16;history.enabled=yes;history.dtHist=0.0;run.nstepmax=8;run.noutput=1000;output.outputVtk=yes;output.outputHdf5=no;output.outputVtkAscii=no", [8]),
    "ot3d_12_history": ("orszag-tang3d", "mesh.nx=12;mesh.ny=12;mesh.nz=12;history.enabled=yes;history.dtHist=0.0;run.nstepmax=5;run.noutput=1000", [5]),
    # --- viscosity and resistivity (operator-split stage after the Godunov update) ----------------------
    "ot2d_24_visc_res": ("orszag-tang", "mesh.nx=24;mesh.ny=24;hydro.nu=0.01;MHD.eta=0.02;run.nstepmax=10;run.noutput=100", [10]),
    "briowu_x_32_res": ("mhd_BrioWu", "mesh.nx=32;mesh.ny=16;BrioWu.direction=0;MHD.eta=0.01;run.nstepmax=10;run.noutput=100", [10]),
    "ot3d_12_visc_res": ("orszag-tang3d", "mesh.nx=12;mesh.ny=12;mesh.nz=12;hydro.nu=0.005;MHD.eta=0.01;run.nstepmax=5;run.noutput=100", [5]),
    "ot3d_12_iso_visc_res": ("orszag-tang3d", "mesh.nx=12;mesh.ny=10;mesh.nz=8;hydro.cIso=0.9;hydro.nu=0.005;MHD.eta=0.01;run.nstepmax=4;run.noutput=100", [4]),
    "mri_8x16x8_visc_res": ("mhd_mri_3d", "mesh.nx=8;mesh.ny=16;mesh.nz=8;MRI.amp=0.2;hydro.nu=1e-6;MHD.eta=2e-6;run.nstepmax=8;run.noutput=1000", [8]),
    "implode3d_12_visc": ("implode3d", "mesh.nx=12;mesh.ny=12;mesh.nz=12;hydro.riemannSolver=hllc;hydro.nu=0.002;run.nstepmax=6;run.noutput=100", [6]),
    "blast2d_24x36_visc": ("blast2d", "mesh.nx=24;mesh.ny=36;hydro.nu=0.001;run.nstepmax=8;run.noutput=100", [8]),
    "sod2d_32x8": ("hydro_sod2d", "mesh.nx=32;mesh.ny=8;run.nstepmax=10;run.noutput=100", [0, 10]),
}

VAR_NAMES = {
    4: ["density", "energy", "mx", "my"],
    5: ["density", "energy", "mx", "my", "mz"],
    8: ["density", "energy", "mx", "my", "mz", "bx", "by", "bz"],
}


def apply_overrides(ini_text, overrides):
    """Return ini_text with 'section.key=value;...' applied (keys are case-insensitive like the reference)."""
    lines = ini_text.splitlines()
    for ov in [o for o in overrides.split(";") if o.strip()]:
        lhs, value = ov.split("=", 1)
        section, key = lhs.strip().split(".", 1)
        out, cur, done, sec_end = [], None, False, None
        for ln in lines:
            m = re.match(r"\s*\[(.+?)\]", ln)
            if m:
                if cur is not None and cur.lower() == section.lower() and not done and sec_end is None:
                    sec_end = len(out)
                cur = m.group(1)
            elif cur is not None and cur.lower() == section.lower():
                km = re.match(r"\s*([^=#;]+?)\s*=", ln)
                if km and km.group(1).lower() == key.lower():
                    ln = "%s=%s" % (key, value.strip())
                    done = True
            out.append(ln)
        if not done:
            if cur is not None and cur.lower() == section.lower() and sec_end is None:
                sec_end = len(out)
            if sec_end is None:
                out += ["", "[%s]" % section, "%s=%s" % (key, value.strip())]
            else:
                out.insert(sec_end, "%s=%s" % (key, value.strip()))
        lines = out
    return "\n".join(lines) + "\n"


def read_vti(path):
    """Parse the reference's hand-written appended-raw VTI; returns dict name -> array[nz][ny][nx]."""
    blob = open(path, "rb").read()
    head_end = blob.index(b'<AppendedData encoding="raw">')
    head = blob[:head_end].decode("ascii", "replace")
    ext = re.search(r'WholeExtent="(\d+) (\d+) (\d+) (\d+) (\d+) (\d+)"', head)
    e = [int(x) for x in ext.groups()]
    nx, ny, nz = e[1] - e[0] + 1, e[3] - e[2] + 1, e[5] - e[4] + 1
    start = blob.index(b"_", head_end) + 1
    out = {}
    for m in re.finditer(r'<DataArray type="Float64" Name="(\w+)" format="appended" offset="(\d+)"', head):
        name, off = m.group(1), int(m.group(2))
        (nbytes,) = struct.unpack_from("<I", blob, start + off)
        assert nbytes == nx * ny * nz * 8, (name, nbytes, nx, ny, nz)
        a = np.frombuffer(blob, dtype="<f8", count=nx * ny * nz, offset=start + off + 4)
        out[name] = a.reshape(nz, ny, nx).copy()
    return out, (nx, ny, nz)


def run_case(name):
    base, overrides, steps = CASES[name]
    ini = open(os.path.join(ROOT, "configs", base + ".ini")).read()
    ini = apply_overrides(ini, overrides + ";run.nlog=1;output.outputVtk=yes;output.outputHdf5=no;output.outputDir=./")
    prefix = re.search(r"outputPrefix=(\S+)", ini).group(1)
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "case.ini"), "w") as f:
            f.write(ini)
        res = subprocess.run([REF_BIN, "--param", "case.ini"], cwd=td, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, universal_newlines=True, check=True)
        log = res.stdout
        arrays = {}
        for s in steps:
            fields, dims = read_vti(os.path.join(td, "%s_%07d.vti" % (prefix, s)))
            names = VAR_NAMES[len(fields)]
            arrays["step_%d" % s] = np.stack([fields[n] for n in names])
        # history file of the run, if it wrote one ([history] enabled=yes): rows of
        # totalTime dt mass maxwell reynolds maxwell+reynolds magp mean_Bx mean_By mean_Bz divB (6 significant digits)
        hist = os.path.join(td, prefix + "_history.txt")
        if os.path.exists(hist):
            lines = [ln.rstrip("\n") for ln in open(hist) if ln.strip() and not ln.startswith("#")]
            try:
                arrays["history"] = np.array([[float(x) for x in ln.split()] for ln in lines])
            except ValueError:
                # history_inertial_wave prints totalTime and dt without a separator: keep the rows as text
                arrays["history_text"] = np.array(lines)
    # dt log: "step=  N t=  T dt=  D" lines; with nlog=1 the line printed at step N carries the dt of step N-1
    # (the first one carries the initial compute_dt).  Duplicated lines (output + log) are collapsed on N.
    dts = {}
    for m in re.finditer(r"step=\s*(\d+)\s+t=\s*([-+0-9.eE]+)\s+dt=\s*([-+0-9.eE]+)", log):
        dts[int(m.group(1))] = (float(m.group(2)), float(m.group(3)))
    nmax = max(dts) if dts else -1
    arrays["log_t"] = np.array([dts[i][0] for i in range(nmax + 1)])
    arrays["log_dt"] = np.array([dts[i][1] for i in range(nmax + 1)])
    tt = re.search(r"DEBUG : totalTime\s+([-+0-9.eE]+)", log)
    # (the hydro run class does not print it; NaN then)
    arrays["total_time"] = np.array(float(tt.group(1)) if tt else float("nan"))
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **arrays)
    return {"base": base, "overrides": overrides, "steps": steps}


def main(argv):
    if not os.path.exists(REF_BIN):
        sys.exit("oracle/_ref/euler_cpu missing: run `make -C oracle -f Makefile.ref` where /root/reference exists")
    names = argv or list(CASES)
    meta_path = os.path.join(GOLDEN, "cases.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    for n in names:
        meta[n] = run_case(n)
        print("golden:", n, meta[n]["steps"])
    with open(meta_path, "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])

echo "== default"; RGPU_CHUNKS=1 python scripts/gpu_probe.py --no-parity 256 2>&1 | grep -A1 "mhd_mri_3d "
for w in 4 5; do echo "== waves=$w"; RGPU_LIB=$PWD/build/librgpu_w$w.so RGPU_CHUNKS=1 python scripts/gpu_probe.py --no-parity 256 2>&1 | grep -A1 "mhd_mri_3d "; done

echo "== exact"; python scripts/gpu_probe.py --no-parity 256 2>&1 | grep -A1 "mhd_mri_3d\|implode"
echo "== fma"; RGPU_LIB=$PWD/build/librgpu_fma.so python scripts/gpu_probe.py 256 2>&1 | grep -v "mismatching=0/" | tail -12

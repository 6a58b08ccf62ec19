for c in 1 2 4 8 16 32; do echo "== RGPU_CHUNKS=$c"; RGPU_CHUNKS=$c python scripts/gpu_probe.py --no-parity ${1:-256} 2>&1 | grep "mhd_mri_3d\|orszag-tang3d" ; done
python scripts/gpu_probe.py 2>&1 | grep "TOTAL mism"

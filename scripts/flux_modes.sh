for m in 0 1 2 3; do echo "== RGPU_FLUX_MODE=$m"; RGPU_FLUX_MODE=$m python scripts/gpu_probe.py --no-parity 256 2>&1 | grep -A1 "mhd_mri_3d"; done

for pr in -1 0 1; do for c in 8 16; do echo "== ALU_PRIO=$pr CHUNKS=$c"; RGPU_ALU_PRIO=$pr RGPU_CHUNKS=$c python scripts/gpu_probe.py --no-parity 256 2>&1 | grep "mhd_mri_3d " ; done; done

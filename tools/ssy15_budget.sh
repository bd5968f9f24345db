for b in 0 1024 4096; do echo "== SDFS_TILE_BUDGET=$b"; SDFS_TILE_BUDGET=$b timeout -k 10 100 python tools/ssy15_breakdown.py 2>&1 | grep -v amdgpu.ids | grep -v "^$"; done

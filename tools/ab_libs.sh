# Developer aid, run on the GPU box from the repository root: the chained-field probes with two libraries, alternating in ONE session
# (x265_amd/libx265amd_old.so against the built one).
for rep in 1 2; do for lib in libx265amd_old.so libx265amd.so; do echo "== $lib"; X265AMD_LIB=$PWD/x265_amd/$lib timeout -k 10 200 python tools/field_multi_probe.py 2>&1 | grep -v amdgpu.ids | grep "8x8 \|16x8\|16x16 mode\|all eight\|all twenty"; done; done

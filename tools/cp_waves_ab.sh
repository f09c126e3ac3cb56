# Developer aid, run on the GPU box from the repository root: the chained field's timings (tools/prof_field.py, tools/field_multi_probe.py)
# with four and with eight wavefronts per workgroup of the candidate-parallel classes (X265AMD_CP_WAVES), alternating in ONE
# session - timings of different boxes differ by a few percent.
for w in 4 8 4 8; do echo "== X265AMD_CP_WAVES=$w"; X265AMD_CP_WAVES=$w timeout -k 10 120 python tools/prof_field.py 2>&1 | grep -v amdgpu | cut -c1-100 | grep -v "method 3"; X265AMD_CP_WAVES=$w timeout -k 10 200 python tools/field_multi_probe.py 2>&1 | grep -v amdgpu.ids | tail -5; done

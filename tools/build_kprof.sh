# dev-only: libv1t_amd_kprof.so = the library with the attention kernels' s_memtime stamps compiled in
set -e
cd "$(dirname "$0")/.."
python -c "import v1t_amd.build as b; b.build(verbose=False)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DV1T_KPROF $V1T_EXTRA -c v1t_amd/csrc/attention.hip -o /tmp/attention_kprof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o v1t_amd/lib/libv1t_amd_kprof.so v1t_amd/lib/api.o v1t_amd/lib/gemm.o /tmp/attention_kprof.o v1t_amd/lib/elementwise.o v1t_amd/lib/readout.o v1t_amd/lib/gridprep.o v1t_amd/lib/metrics.o v1t_amd/lib/data.o

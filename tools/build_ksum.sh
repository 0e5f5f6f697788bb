# dev-only: libv1t_amd_ksum.so = the library with the per-segment cycle sums of attn_fwd2_kernel compiled in
set -e
cd "$(dirname "$0")/.."
python -c "import v1t_amd.build as b; b.build(verbose=False)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-slp-vectorize -DV1T_KSUM $V1T_EXTRA -c v1t_amd/csrc/attention.hip -o /tmp/attention_ksum.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o v1t_amd/lib/libv1t_amd_ksum.so v1t_amd/lib/api.o v1t_amd/lib/gemm.o /tmp/attention_ksum.o v1t_amd/lib/elementwise.o v1t_amd/lib/readout.o v1t_amd/lib/gridprep.o v1t_amd/lib/metrics.o v1t_amd/lib/data.o

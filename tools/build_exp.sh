# dev-only: libv1t_amd_${V1T_EXP_NAME:-exp}.so = the library with gemm.hip compiled with $V1T_EXTRA (experiments; results may be wrong)
set -e
cd "$(dirname "$0")/.."
python -c "import v1t_amd.build as b; b.build(verbose=False)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $V1T_EXTRA -c v1t_amd/csrc/gemm.hip -o /tmp/gemm_exp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o v1t_amd/lib/libv1t_amd_${V1T_EXP_NAME:-exp}.so v1t_amd/lib/api.o /tmp/gemm_exp.o v1t_amd/lib/attention.o v1t_amd/lib/elementwise.o v1t_amd/lib/readout.o v1t_amd/lib/gridprep.o v1t_amd/lib/metrics.o v1t_amd/lib/data.o

for w in 512 768 1024 1536; do echo "== V1T_TN_WGS=$w"; V1T_TN_WGS=$w bash tools/kstat.sh "gemm_tn2|tn_reduce" 2>&1 | tail -5 | cut -c1-140; done

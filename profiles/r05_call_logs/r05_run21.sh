# round-5 GPU call 21: training sanity on the three step paths
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python tools/train_sanity.py 300 2>/dev/null | tail -8 > $O/r05_train_sanity.txt
echo "--- the reference's loop over the registry modules, torch.optim.AdamW" >> $O/r05_train_sanity.txt
python tools/train_sanity_module.py 300 2>/dev/null | tail -8 >> $O/r05_train_sanity.txt
echo "--- the same loop, v1t_amd.FusedAdamW.for_model" >> $O/r05_train_sanity.txt
python tools/train_sanity_module.py 300 fused 2>/dev/null | tail -8 >> $O/r05_train_sanity.txt
cat $O/r05_train_sanity.txt

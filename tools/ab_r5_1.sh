cd $GRAFT_REPO_ROOT
python tools/gpu_quick.py deep_all_s deep_all_big deep_rej256 deep_rej128_s deep_acc100 deep_aa deep_aa_big hard_acc0 hard_rej0 hard_big 2>&1 | tail -40

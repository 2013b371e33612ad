cd $GRAFT_REPO_ROOT
python tools/align_split.py 2>&1 | cut -c1-220
for v in clk1 clk2 clk4; do echo == $v; UGS_LIB=usearch12_amd/variants/libugs_$v.so python tools/align_split.py 2>&1 | cut -c1-300; done

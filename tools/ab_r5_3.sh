cd $GRAFT_REPO_ROOT
for v in clk1 clk2 clk3; do echo == $v; UGS_LIB=usearch12_amd/variants/libugs_$v.so python tools/align_split.py 2>&1 | grep -v "launch:" | sed 's/.*| align:/align:/' | cut -c1-300; done
for v in clk1 clk2 clk3; do echo == aa $v; RQ_SHAPE=aa UGS_LIB=usearch12_amd/variants/libugs_$v.so python tools/align_split.py 2>&1 | grep -v "launch:" | sed 's/.*| align:/align:/' | cut -c1-300; done

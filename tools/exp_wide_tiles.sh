# k_place_wide's tiles per workgroup on config 2 (GAT_PLACE_WIDE=1 forces the wide kernel): tuning builds of the library are expected at
# build/w<N>/libgat_w<N>.so (make -C gat_amd/csrc EXTRA=-DGAT_PLACE_WIDE_TILES=<N> BUILD=$PWD/build/w<N> OUT=$PWD/build/w<N>/libgat_w<N>.so).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "default (lean k_place_pipe)"; python tools/place_scaling.py config2 1250 10000 20000 2>&1 | tail -3
for W in w2 w4; do echo "k_place_wide, $W tiles per workgroup"; GAT_PLACE_WIDE=1 GAT_LIB_PATH=$PWD/build/$W/libgat_$W.so python tools/place_scaling.py config2 1250 10000 20000 2>&1 | tail -3; done
echo "k_place_wide, 8"; GAT_PLACE_WIDE=1 python tools/place_scaling.py config2 1250 10000 20000 2>&1 | tail -3

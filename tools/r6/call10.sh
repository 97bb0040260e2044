export SAR_GRAPH_SPLIT3=1 SAR_GRAPH_SPLIT2=0
G2_LAYERS=2,6 G2_BITS="0 2 4 6 8 32 40 46" tools/g2_ablate.sh run
echo "=== producers at s_setprio 2"
cp tools/bin/libsar_g3_prio2.so tools/bin/libsar_g2_ablate.so
G2_LAYERS=2,6 G2_BITS="0" tools/g2_ablate.sh run

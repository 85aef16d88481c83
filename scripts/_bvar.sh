run() { v=$1; p=$2; echo "== $v parts $p"; AMM_LIB=$PWD/atomsmm_amd/exp/lib_$v.so bash scripts/kstats_probe.sh r3n_$v$p --cluster 1 --option build_parts=$p | grep -E "world"; python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/r3n_$v$p'+'_stats/**/*kernel_trace.csv',recursive=True)[0]
v=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'k_cbuild<false' in r['Kernel_Name']]
big=[x for x in v if x>0.5*max(v)]
print('   rebuild kernel: n %d avg %.1f us'%(len(big),sum(big)/len(big)))
PY
}
run b6q192 2; run b6q192 4; run b5q192 3; run b5q192 4; run b4q192 4; run b4q192 5; run b4q192 6; run b3q192 4; run b3q192 6; run b3q192 8

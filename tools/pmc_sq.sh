export TMPDIR=/tmp
root=$PWD
out=gpurun_out/pmc2
rm -rf $out; mkdir -p $out
cd /tmp
for set in "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
   name=$(echo $set | tr ' ' '+' | cut -c1-30)
   rocprofv3 --pmc $set --output-format csv -d $root/$out/pmc_$name -o p -- python3 $root/bench.py --lattice 100 --steps 4 --warmup 2 --no-cpu > $root/$out/log_$name.txt 2>&1
done
cd $root
python3 tools/summarize_profile.py $out

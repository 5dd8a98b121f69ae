R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wg; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/tune_wgrad.py --batch 64 --layers L2p,L3p,L4,L5r --iters 3"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- $CMD > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $O/pmc_mfma -o p --output-format csv -- $CMD > /dev/null 2> $O/pmc_mfma.err
python $R/tools/parse_pmc.py $O/wg_pmc.json $O/pmc_fetch $O/pmc_mfma > $O/wg_pmc.txt 2>&1
grep -i "wgrad" $O/wg_pmc.txt | cut -c1-330
rm -rf $O/pmc_fetch $O/pmc_mfma

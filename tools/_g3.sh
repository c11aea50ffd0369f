mkdir -p gpurun_out/r04b
FQTK_DIRECT_SHAPE=thin tools/pmc_table2.sh r04b/pmc_cfg5_thin "--config 5" > gpurun_out/r04b/pmc2_cfg5_thin.txt 2>&1
FQTK_DIRECT_SHAPE=fat4 tools/pmc_table2.sh r04b/pmc_cfg5_fat4 "--config 5" > gpurun_out/r04b/pmc2_cfg5_fat4.txt 2>&1
tools/pmc_table2.sh r04b/pmc_cfg3_lds "--config 3 --reads 100000000" > gpurun_out/r04b/pmc2_cfg3_lds.txt 2>&1
tools/pmc_table2.sh r04b/pmc_cfg3_tab "--config 3 --reads 100000000 --memo-table" > gpurun_out/r04b/pmc2_cfg3_tab.txt 2>&1
grep -h "r04b" gpurun_out/r04b/pmc2_*.txt

# usage: bash tools/gpu_pmc.sh <outdir> "<counters>" <kernel name filters...>
O=$1; shift; CNT=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-baseline 0 --graph 0 --preroll 70 --warmup 2 --steps 20 > $GRAFT_REPO_ROOT/$O/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc "$@" > $O/pmc_summary_$(echo $CNT | tr ' ' '_' | cut -c1-40).txt 2>&1
rm -rf $O/pmc
cat $O/pmc_summary_*.txt

# HBM traffic of apply_emb alone (gather_rows_kernel) by precision at one batch size: two separate --pmc passes each
# usage (on the GPU box): bash tools/gr_pmc.sh <tag> <B> <bits ...>
TAG=$1; B=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for bits in "$@"; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-include-regex "evs::gather_rows" --output-format csv -d $OUT/u$bits/rd -- python3 $ROOT/tools/gather_bench.py $bits B=$B > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-include-regex "evs::gather_rows" --output-format csv -d $OUT/u$bits/wr -- python3 $ROOT/tools/gather_bench.py $bits B=$B > /dev/null 2>&1
  echo "== u$bits B=$B"; python3 $ROOT/tools/pmc_summary.py $OUT/u$bits "evs::"
done
find $OUT -name "*.db" -delete

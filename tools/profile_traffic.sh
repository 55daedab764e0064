#!/bin/bash
# Pin the force kernel's memory traffic: the L2's fabric-side read requests BY SIZE
# (TCC_EA0_RDREQ_32B / _64B / _128B -- bytes = 32 n32 + 64 n64 + 128 n128, no 1x / 2x guess), writes, DRAM-bound
# requests and L2 hit / miss, for (a) the calibration kernels of known traffic, (b) the product run, (c) the
# diagnostics build's force kernel with nothing / the field-map sampling / the pair work switched off: the
# differences are the map bytes and the neighbour-gather bytes.    bash tools/profile_traffic.sh TAG [bench args]
TAG=${1:?tag}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/traffic_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
P1="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
P2="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_ATOMIC_sum"
hipcc -O3 --offload-arch=gfx950 -o "$OUT/gather_traffic" tools/microbench/gather_traffic.hip || exit 1
cd /tmp
pass() {   # name, counters, command...
    local name=$1 ctr=$2; shift 2
    rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d "$OUT/$name" -- "$@" > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
    echo "pass $name done"
}
pass cal_p1 "$P1" "$OUT/gather_traffic"
pass cal_p2 "$P2" "$OUT/gather_traffic"
B="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile --no-fast-leg $*"
for p in 1 2 3; do eval ctr=\$P$p; pass product_p$p "$ctr" $B; done
pass product_fetch "FETCH_SIZE" $B
pass product_write "WRITE_SIZE" $B
export PEDONI_HIP_LIB=$ROOT/pedoni_amd/lib/libpedoni_hip_diag.so
for v in 256 291 260 258 257 288; do      # nothing off; no map sampling at all (1|2|32); no pairs (4); no wall term (2); no goal stencil (1); no despawn sample (32)
    export PEDONI_ABLATE=$v
    pass abl${v}_p1 "$P1" $B
    pass abl${v}_p2 "$P2" $B
done
unset PEDONI_ABLATE PEDONI_HIP_LIB
cd "$ROOT"
python3 tools/traffic_summary.py "$OUT" "$TAG"

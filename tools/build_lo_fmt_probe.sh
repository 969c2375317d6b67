# Round 6, rate experiment: what would the lo products cost on the FP6 / FP4 forms of v_mfma_scale_f32_16x16x128_f8f6f4?
# Builds eventclip_amd/libeventclip_hip_lofmt$1.so = the diagnostic library with the e4m3 segments' products issued with
# operand format code $1 (2 = e2m3 / FP6, 4 = e2m1 / FP4) ON THE SAME BYTES AND THE SAME STAGING: results are meaningless,
# the K-tile cadence is what an FP6 / FP4 lo product padded to the e4m3 row pitch would run at.
#   bash tools/build_lo_fmt_probe.sh 2 && EVENTCLIP_HIP_LIB=eventclip_amd/libeventclip_hip_lofmt2.so python tools/bench_gemm_fp8.py
set -e
fmt=${1:-2}
cd "$(dirname "$0")/.."
python -m eventclip_amd.build --diag > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-gpu-rdc -Wall -Wno-unused-function -I include \
    -DEC_GEMM_DIAG -DEC_ATTN_DIAG -DEC_EVENTS_DIAG -DEC_LO_FMT=$fmt -c eventclip_amd/csrc/gemm.hip -o eventclip_amd/csrc/gemm.lofmt$fmt.o
objs=$(ls eventclip_amd/csrc/*.diag.o | grep -v /gemm.diag.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o eventclip_amd/libeventclip_hip_lofmt$fmt.so $objs eventclip_amd/csrc/gemm.lofmt$fmt.o
echo eventclip_amd/libeventclip_hip_lofmt$fmt.so

# On the GPU box: tools/stride_probe.py for u8 / u16 / u4 at the tight stride (shipped library) and at 64 / 128 / 32 bytes
# (libraries built by tools/variants.sh s64@evs_fused_rfq:"-DEVS_X_STRIDE=64" ...)
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$R/ev-store-dlrm_amd/lib/var
for rep in 1 2; do
python3 tools/stride_probe.py 8 36 2>/dev/null
EVS_LIB_PATH=$V/libevstore_hip_s64.so python3 tools/stride_probe.py 8 64 2>/dev/null
python3 tools/stride_probe.py 16 72 2>/dev/null
EVS_LIB_PATH=$V/libevstore_hip_s128.so python3 tools/stride_probe.py 16 128 2>/dev/null
python3 tools/stride_probe.py 4 18 2>/dev/null
EVS_LIB_PATH=$V/libevstore_hip_s32.so python3 tools/stride_probe.py 4 32 2>/dev/null
done

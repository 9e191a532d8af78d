# usage: bash scripts/_pan_dev.sh "<extra -D flags>"   (development: one instantiation, then stamps + probe on the GPU)
set -e
cd /root/repo/recad_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wall -Wno-unused-function -DPAN_DEV $1 -c score_topk.hip -o score_topk.o 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/librecad_hip.so *.o
cd /root/repo
timeout 1500 gpurun --timeout 600 -- 'bash scripts/_pan_run2.sh' 2>&1 | grep -vE "^\[gpurun\] (sending|status)|amdgpu.ids"

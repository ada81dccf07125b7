#!/bin/bash
# register / LDS / scratch use of every kernel of frx_device.hip (device-only compile; no GPU needed)
cd "$(dirname "$0")/../fast-racing_amd/csrc"
for tu in frx_device frx_device_round frx_device_eval frx_device_solo; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only --no-gpu-bundle-output -c -o /tmp/$tu.o $tu.hip "$@" || exit 1
done
for tu in frx_device frx_device_round frx_device_eval frx_device_solo; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/$tu.o; done | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk).group(1)
    g=lambda k: re.search(r'\.'+k+r':\s+(\S+)',blk).group(1)
    print(name[:70].ljust(70), 'agpr',blk.split()[0],'vgpr',g('vgpr_count'),'sgpr',g('sgpr_count'),'spill',g('vgpr_spill_count'),'scratch',g('private_segment_fixed_size'))
"

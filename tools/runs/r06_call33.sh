#!/bin/bash
# round 6, GPU call 33: phase stamps of the decode attention at B = 8 (probe build)
cd $GRAFT_REPO_ROOT
export AFK_LIB_PATH=$GRAFT_REPO_ROOT/audio-flamingo_amd/lib_probes/libafk.so
show() { python -c "
import sys, json
txt = sys.stdin.read(); d = json.loads(txt[txt.index('{'):])
print({k: v for k, v in d.items() if k != 'launches'})
for r in d['launches'][:3]: print(r)
"; }
python tools/probes/probe_attn_decode.py 800 8 2>/dev/null | show
NS=2 python tools/probes/probe_attn_decode.py 800 8 2>/dev/null | show
python tools/probes/probe_attn_decode.py 800 1 2>/dev/null | show
python tools/probes/probe_attn_decode.py 200 8 2>/dev/null | show

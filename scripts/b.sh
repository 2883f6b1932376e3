#!/bin/bash
# quiet in-tree build: prints compiler diagnostics only
cd "$(dirname "$0")/.." && timeout 1500 python -c "from nmf_toolbox_amd import build; build.build()" 2>&1 | grep -E "error:|warning:|undefined reference|multiple definition|Error" | head -30; ls -la nmf_toolbox_amd/libnmfx.so | awk '{print $6,$7,$8}'

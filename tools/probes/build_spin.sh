#!/bin/bash
cd "$(dirname "$0")" && for f in spin mfma_peak; do "${HIPCC:-/opt/rocm/bin/hipcc}" --offload-arch=gfx950 -O2 -shared -fPIC $f.hip -o lib$f.so && echo built lib$f.so; done

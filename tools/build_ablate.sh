# usage (build container): bash tools/build_ablate.sh 0 1 2 3 4   -> tools/ab/lib_abl<n>.so (diagnostic builds of
# libpalace_hip.so with -DPALACE_ABL=<n>; see the PALACE_ABL switches in palace_amd/csrc/eref.hip)
cd "$(dirname "$0")/.." || exit 1
mkdir -p tools/ab /tmp/abl
for n in "$@"; do
  for f in ctx eref graph match depth; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPALACE_ABL=$n -c palace_amd/csrc/$f.hip -o /tmp/abl/$f.$n.o &
  done
  wait
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/ab/lib_abl$n.so /tmp/abl/ctx.$n.o /tmp/abl/eref.$n.o /tmp/abl/graph.$n.o /tmp/abl/match.$n.o /tmp/abl/depth.$n.o
done
ls -la tools/ab

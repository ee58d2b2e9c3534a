# usage (build container): bash tools/build_variant.sh <name> [-DMACRO=value ...]   -> tools/ab/lib_<name>.so
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
mkdir -p tools/ab /tmp/var_$name
for f in ctx eref eref_scan eref_index eref_table graph match decomp filter depth inflate; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c palace_amd/csrc/$f.hip -o /tmp/var_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/ab/lib_$name.so /tmp/var_$name/*.o && ls -la tools/ab/lib_$name.so

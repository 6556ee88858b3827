# Diagnostic: the same tool against two builds of the library on ONE box, alternating: bash tools/lib_ab.sh <libB.so> <tool.py> [args]
B=$1; shift
for i in 1 2; do
  echo "--- product build"; python "$@" 2>&1 | tail -${TAILN:-3}
  echo "--- $B"; EDADM_LIB_PATH=$B python "$@" 2>&1 | tail -${TAILN:-3}
done

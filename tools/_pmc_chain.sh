# PMC passes of the vocoder's chain kernels for two builds (tree's = prefetch form, lib/alt/libbisinger_pf0.so = first form)
for lib in "" "$PWD/bisinger_amd/lib/alt/libbisinger_pf0.so"; do
  echo "==== BSG_LIB=$lib"
  export BSG_LIB=$lib
  [ -z "$lib" ] && unset BSG_LIB
  PROG=tools/prof_vocoder.py FL_ONLY=resblock_chain PB=16 PN=1 bash tools/_fl_pmc.sh
done

#!/bin/bash
# kernel traces of train-on-forces steps (tools/force_trace.py, modes train / train-reference) -> the kernels that are not
# this library's (tools/aten_in_trace.py), gathered into gpurun_out/force_train_non_library_kernels.txt
#   bash tools/gpu_force_train_trace.sh <commit> [mols] [steps]
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD
COMMIT=${1:-unknown}; MOLS=${2:-256}; STEPS=${3:-6}
OUT=$REPO/gpurun_out/force_train_non_library_kernels.txt
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
{
echo "# Kernels that are NOT this library's in rocprofv3 --kernel-trace runs of train-on-forces steps (finetune_md17.py:30-54;"
echo "# tools/force_trace.py <backbone> $MOLS $STEPS <mode>, tools/aten_in_trace.py) @ $COMMIT"
echo "# mode train: the step on the library's pieces (Dense head, ops.energy_force_loss, gradients added in place, fused Adam)"
echo "# mode train-reference: the reference's lines as written (torch L1Loss arithmetic, loss.backward(), torch.optim.Adam)"
} > $OUT
for which in schnet painn; do
  for mode in train train-reference; do
    D=/tmp/ft_${which}_${mode}
    rm -rf $D
    ( cd /tmp && rocprofv3 --kernel-trace --stats -d $D -- python3 $REPO/tools/force_trace.py $which $MOLS $STEPS $mode ) > /tmp/ft.log 2>&1
    echo >> $OUT
    echo "## python3 tools/force_trace.py $which $MOLS $STEPS $mode" >> $OUT
    grep "ms per step" /tmp/ft.log >> $OUT || tail -5 /tmp/ft.log >> $OUT
    python3 $REPO/tools/aten_in_trace.py $D >> $OUT 2>&1
  done
done
cat $OUT

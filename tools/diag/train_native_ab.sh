mkdir -p gpurun_out/r05f
export MIOPEN_FIND_MODE=3
for rep in 1 2; do
for nat in 0 1; do
  GHN3_NATIVE_OPS=$nat timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step|Error|error" | tail -2 | sed "s/^/native=$nat pass=$rep: /" | tee -a gpurun_out/r05f/train_ab.txt
done
done

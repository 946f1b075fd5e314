cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rate; python -c "
from casmvsnet_pl_amd import ops
names = {1: '16x16x4', 4: '16x16x4+ds_read'}
for shape in (1, 4):
    for blocks in (256, 512, 768, 1024, 2048):
        print('mfma', names[shape], 'blocks', blocks, 'TFLOP/s %.1f' % ops.selftest_mfma_rate(shape, blocks, 4096))
" 2>&1 | tee gpurun_out/rate/rate.txt

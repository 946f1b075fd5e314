cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rate; python -c "
from casmvsnet_pl_amd import ops
names = {4: '16x16x4+ds_read linear', 5: '+PX pattern (2j+u), 32 KiB', 6: '+PX pattern, 42 KiB LDS (3 wg/CU)', 7: '+random data'}
for shape in (4, 5, 6, 7):
    for blocks in (256, 768, 1536):
        print('mfma', names[shape], 'blocks', blocks, 'TFLOP/s %.1f' % ops.selftest_mfma_rate(shape, blocks, 4096))
" 2>&1 | tee gpurun_out/rate/rate2.txt

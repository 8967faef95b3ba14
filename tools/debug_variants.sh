run() { echo "== $1"; shift; env "$@" timeout 100 python -c "
import sys, torch; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import os
import test_gpu_train_graph as t
if os.environ.get('SIDE') == '1':
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        t.test_graph_step_equals_eager_step(True, False)
else:
    t.test_graph_step_equals_eager_step(True, False)
torch.cuda.synchronize(); print('variant ok')" 2>&1 | grep -v amdgpu.ids | tail -2; }
run packet_capture_off DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run side_stream SIDE=1
run serialize AMD_SERIALIZE_KERNEL=3
run baseline X=1

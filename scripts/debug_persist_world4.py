"""Debug aid: config 4 at full size, 4 ranks sharing one GPU, no switches -- prints what the persistent kernel's trial reports."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'exp-trmf-nips16_amd'))

if __name__ == '__main__':
    import dist_worker
    import test_dist
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    env = {'TRMF_P2P_VERBOSE': '1'}
    for kv in sys.argv[2:]:
        key, val = kv.split('=')
        env[key] = val
    out = dict(test_dist._spawn(dist_worker.gpu_host_staged, world, 2, 'c3full', env, ('float32',)))
    print(env, out[0]['float32'][6])

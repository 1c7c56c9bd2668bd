#!/usr/bin/env python3
"""Concurrency proxy for the time-slice-chain design: K independent engines of minibatch 256 / K, each
driven from its own torch stream, one update-step each per round.  Not the chain design itself (BatchNorm groups of B / K rows
instead of B) -- it answers the question the design rests on: does running K dependent launch chains CONCURRENTLY hide the
pixel-independent floor of the step (t(B) = ~5 ms + 0.043 ms x B)?  Stored-action loss (graph-capturable).

usage: chain_proxy.py K [steps]   (environment: CDRL_GRAPH, CDRL_SIDE_STREAM as for the engine)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    K = int(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    total = int(os.environ.get('PROXY_TOTAL_B', '256'))
    import torch
    from carla_driving_rl_agent_amd import synthetic
    from carla_driving_rl_agent_amd.engine import LearnerEngine, gae_returns
    from carla_driving_rl_agent_amd.init import init_engine_parameters
    dev = 'cuda:0'
    B = total // K
    engs, batches, streams = [], [], []
    for k in range(K):
        e = LearnerEngine(B, device=dev, T=4, H=90, W=120, compute=os.environ.get('PROXY_DTYPE', 'f32'))
        init_engine_parameters(e, seed=42)
        r = synthetic.make_rollout(B, T=4, H=90, W=120, seed=42 + k)
        states = {n: torch.as_tensor(v).to(dev) for n, v in r['states'].items()}
        rewards = torch.cat([torch.as_tensor(r['reward']).to(dev), torch.zeros(1, device=dev)])
        values = torch.cat([torch.as_tensor(r['value']).to(dev), torch.zeros((1, 2), device=dev)])
        _, returns_be, _, adv = gae_returns(rewards, values, 0.9999, 0.999, 2.0)
        speed = (torch.as_tensor(r['speed'][:, 0]) / 100.0).to(dev).contiguous()
        sim = torch.as_tensor(r['similarity'][:, 0]).to(dev).contiguous()
        pol = dict(states=states, advantages=adv.contiguous(), old_log_prob=torch.as_tensor(r['old_log_prob']).to(dev), speed=speed,
                   similarity=sim, u=torch.as_tensor(r['action']).to(dev), du_da=None, du_db=None)
        val = dict(states=states, returns=returns_be.contiguous(), speed=speed, similarity=sim)
        engs.append(e)
        batches.append((pol, val))
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()

    def round_():
        # interleave the K engines pass by pass so that the host feeds all queues evenly
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                engs[k].policy_forward_backward(batches[k][0])
                engs[k].policy_apply()
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                engs[k].value_forward_backward(batches[k][1])
                engs[k].value_apply()

    for _ in range(4):
        round_()
    torch.cuda.synchronize()
    t0 = time.time()
    host = None
    for i in range(steps):
        round_()
        if i == 3:
            host = (time.time() - t0) / 4
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    print(json.dumps(dict(K=K, B_each=B, ms_per_round=round(ms, 3), host_enqueue_ms=round(host * 1e3, 3),
                          graph=os.environ.get('CDRL_GRAPH', '0'), side=os.environ.get('CDRL_SIDE_STREAM', '1'),
                          loss=engs[0].metrics('policy')['loss'])))


if __name__ == '__main__':
    main()

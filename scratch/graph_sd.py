import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
runner, _, (videos, ids, mask) = ge.make_tiny_setup(torch.float32, seed=3)
blocks = runner.video_encoder.model.blocks
for i, b in enumerate(blocks): b.sd_prob = 0.5 * i / (len(blocks) - 1)
runner._use_step_graph = True
for i in range(9):
    m, _ = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
    print(i, round(m["loss"], 5), "captured" if any(s["graph"] is not None for s in runner._graphs.values()) else "eager", flush=True)

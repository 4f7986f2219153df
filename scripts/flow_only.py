"""One S3Gen flow + HiFT pass at the bench shape (B=8, 250 tokens, 10 s prompt) -- no hipGraph: used for PMC collection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatterbox_amd import synth
from chatterbox_amd.hift import HiFTEngine
from chatterbox_amd.s3gen import FlowEngine
dev = torch.device("cuda:0")
sd = synth.s3gen_state_dict(0)
flow, hift = FlowEngine(sd, dev), HiFTEngine(sd, dev)
B, N = 8, 250
toks = torch.stack([synth.speech_tokens(N, seed=b) for b in range(B)]).to(dev)
lens = torch.full((B,), N, dtype=torch.int32, device=dev)
for _ in range(int(os.environ.get("CBX_REPS", "1"))):
    mel = flow.inference(toks, lens, synth.s3gen_ref())
    wav, _ = hift.inference(mel)
torch.cuda.synchronize()
print("ok", wav.shape)

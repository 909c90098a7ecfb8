import torch, time
dev="cuda"
K,V=10,74053
logits=torch.randn(K,V,device=dev).bfloat16()
beam=torch.zeros(1,K,device=dev)
def part():
    logp=torch.log_softmax(logits.float(),-1)
    scores=(logp+beam.view(-1,1)).view(1,K*V)
    s,i=torch.topk(scores,2*K,dim=1,largest=True,sorted=True)
    return s.tolist(), i.tolist()
for _ in range(5): part()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(100): part()
torch.cuda.synchronize(); print("log_softmax + add + topk + 2 tolist: %.1f us" % ((time.perf_counter()-t0)/100*1e6))
def p2():
    logp=torch.log_softmax(logits.float(),-1)
    scores=(logp+beam.view(-1,1)).view(1,K*V)
    return scores
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(100): p2()
torch.cuda.synchronize(); print("log_softmax + add: %.1f us" % ((time.perf_counter()-t0)/100*1e6))
sc=p2()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(100): torch.topk(sc,2*K,dim=1)
torch.cuda.synchronize(); print("topk: %.1f us" % ((time.perf_counter()-t0)/100*1e6))
# two-stage topk: per row top-2K then over K*2K
def tk2():
    s1,i1=torch.topk(sc.view(K,V),2*K,dim=1)
    s2,j=torch.topk(s1.view(1,-1),2*K,dim=1)
    return s2, j
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(100): tk2()
torch.cuda.synchronize(); print("two-stage topk: %.1f us" % ((time.perf_counter()-t0)/100*1e6))

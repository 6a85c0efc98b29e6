import torch, time
M=32768
for K,N in [(1024,4096),(4096,4096)]:
    a=torch.randn(M,K,device='cuda'); w=torch.randn(K,N,device='cuda'); b=torch.randn(N,device='cuda')
    for _ in range(3): o=torch.relu(torch.addmm(b,a,w))
    torch.cuda.synchronize(); t=time.time()
    for _ in range(10): o=torch.relu(torch.addmm(b,a,w))
    torch.cuda.synchronize(); dt=(time.time()-t)/10
    print(K,N,"ms",dt*1e3,"TF",2*M*K*N/dt/1e12)

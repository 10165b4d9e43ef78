import torch
dev=torch.device("cuda")
for (M,N,K) in [(263168,512,2048),(32768,512,3072),(263168,512,4096)]:
    a=torch.randn(M,K,device=dev,dtype=torch.bfloat16); b=torch.randn(N,K,device=dev,dtype=torch.bfloat16)
    for _ in range(5): c=a@b.t()
torch.cuda.synchronize()

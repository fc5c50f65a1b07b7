import ctypes, torch
from tricolo_amd import _C
lib=_C.lib()
D=_C.TriConvDesc
torch.zeros(1,device='cuda')
for hw,c in [(32,64),(16,128),(8,256),(4,512)]:
    d=D(192,1,hw,hw,c,1,hw,hw,c,1,3,3,1,0,1,1)
    print(hw,c, [hex(lib.tri_conv_kernel_family(ctypes.byref(d),t,2)) for t in (0,1)], lib.tri_conv_num_mtiles(ctypes.byref(d),2), flush=True)

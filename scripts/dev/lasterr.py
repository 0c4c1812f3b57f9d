import ctypes as C, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
hip=C.CDLL('libamdhip64.so')
hip.hipGetErrorString.restype=C.c_char_p
def chk(tag):
    e=hip.hipGetLastError()
    print(tag, e, hip.hipGetErrorString(e).decode())
from vdlm2dec_amd.demod import Receiver, plan_channels
import scenarios as S
chk('start')
rx=Receiver(2_000_000, plan_channels(S.FC, [-50000]), fmt="cu8", max_push=4096)
chk('created')
got=rx.decode_blocks([(1,10,bytes(2040))])
chk('decoded')
rx.close()
chk('closed')
rx=Receiver(2_000_000, plan_channels(S.FC, [-50000]), fmt="cu8", max_push=4096)
chk('created2')

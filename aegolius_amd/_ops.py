"""Opcode table of the sdfk register machine, parsed from csrc/sdfk_ops.def (the single source of
truth shared with the HIP interpreter kernel and the hiprtc code generator)."""
import os
import re

_DEF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "sdfk_ops.def")
_RX = re.compile(r"^SDFK_OP\(\s*(\w+)\s*,\s*(\w+)\s*,\s*(-?\d+)\s*,\s*(\w+)\s*\)", re.M)

KINDS = ("C_C", "V_C", "V_V", "V_VV")


class OpInfo:
    __slots__ = ("code", "name", "kind", "nparams", "func")

    def __init__(self, code, name, kind, nparams, func):
        self.code, self.name, self.kind, self.nparams, self.func = code, name, kind, nparams, func

    def __repr__(self):
        return "OpInfo(%d, %s, %s, %d)" % (self.code, self.name, self.kind, self.nparams)


def _parse():
    with open(_DEF) as f:
        text = f.read()
    ops = []
    for i, m in enumerate(_RX.finditer(text)):
        name, kind, nparams, func = m.group(1), m.group(2), int(m.group(3)), m.group(4)
        if kind not in KINDS:
            raise ValueError("sdfk_ops.def: unknown kind %r for %s" % (kind, name))
        ops.append(OpInfo(i, name, kind, nparams, func))
    if not ops or len(ops) > 255:
        raise ValueError("sdfk_ops.def: %d opcodes parsed" % len(ops))
    return ops


OPS = _parse()
BY_NAME = {o.name: o for o in OPS}

# register-file limits of the interpreter kernel (csrc/sdfk.hip SDFK_NC / SDFK_NV)
INTERP_NC = 8
INTERP_NV = 8

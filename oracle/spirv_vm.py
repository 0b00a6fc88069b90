"""A small SPIR-V interpreter — TEST INFRASTRUCTURE, used only by tests/golden/make_spirv_vectors.py.

Why it exists: the reference cannot be built or run in this image (Rust + Vulkan), but it ships the COMPILED compute
shaders of the cull path (shaders/*.comp.spv, shaderc over glslang, spirv-opt -O).  Those binaries are reference-held
statements of the algorithm that can be executed without a Vulkan driver: this module runs one of them on the CPU,
invocation by invocation in SIMT lock step (numpy across the lanes of a workgroup), against buffers laid out exactly
as the shader's own Offset / ArrayStride / MatrixStride decorations say.  The outputs pin oracle/orbit_oracle.c: they
are what the reference's shader code computes, not what a second reading of its GLSL says.

What SPIR-V leaves to the implementation is decided here the way the oracle documents it ("canonical
arithmetic"), and nowhere else:
  * every FAdd / FSub / FMul / FDiv is one IEEE binary32 operation, Fma is fused (exactly rounded), Sqrt is correctly
    rounded; nothing is contracted or reassociated that the binary does not contract itself;
  * Dot, MatrixTimesVector, MatrixTimesMatrix, Length sum their products left to right, each product and each sum
    rounded (((a0*b0 + a1*b1) + a2*b2) + a3*b3);
  * Log2 is the oracle's software log2 (passed in by the caller);
  * image sampling with the ReduceMin sampler is the oracle's texel-footprint model (passed in by the caller);
  * atomics and subgroup operations see the invocations in index order (the reference's append order is
    nondeterministic; canonical order = invocation order), subgroup size is a parameter.
Only the opcodes the six shaders of the path use are implemented; anything else raises.

Nothing under orbit_amd/ imports this file, and nothing at test time reads /root/reference: the vectors are generated
once, here, and committed as data under tests/golden/."""
import struct

import numpy as np

F32, U32, I32 = np.float32, np.uint32, np.int32

# ------------------------------------------------------------------------------------------------ module parsing
OP = dict(
    Undef=1, Name=5, MemberName=6, ExtInstImport=11, ExtInst=12, EntryPoint=15, ExecutionMode=16, TypeVoid=19,
    TypeBool=20, TypeInt=21, TypeFloat=22, TypeVector=23, TypeMatrix=24, TypeImage=25, TypeSampler=26,
    TypeSampledImage=27, TypeArray=28, TypeRuntimeArray=29, TypeStruct=30, TypePointer=32, TypeFunction=33,
    ConstantTrue=41, ConstantFalse=42, Constant=43, ConstantComposite=44, ConstantNull=46, SpecConstant=50,
    SpecConstantOp=52, Function=54, FunctionEnd=56, Variable=59, Load=61, Store=62, AccessChain=65,
    InBoundsAccessChain=66, Decorate=71, MemberDecorate=72, VectorShuffle=79, CompositeConstruct=80,
    CompositeExtract=81, CompositeInsert=82, CopyObject=83, SampledImage=86, ImageSampleExplicitLod=88, ImageFetch=95,
    ImageWrite=99, Image=100, ImageQuerySizeLod=103, ImageQuerySize=104, ConvertFToU=109, ConvertFToS=110,
    ConvertSToF=111, ConvertUToF=112, UConvert=113, SConvert=114, Bitcast=124, SNegate=126, FNegate=127, IAdd=128,
    FAdd=129, ISub=130, FSub=131, IMul=132, FMul=133, UDiv=134, SDiv=135, FDiv=136, UMod=137, VectorTimesScalar=142,
    MatrixTimesVector=145, MatrixTimesMatrix=146, Dot=148, LogicalOr=166, LogicalAnd=167, LogicalNot=168, Select=169,
    IEqual=170, INotEqual=171, UGreaterThan=172, SGreaterThan=173, UGreaterThanEqual=174, SGreaterThanEqual=175,
    ULessThan=176, SLessThan=177, ULessThanEqual=178, SLessThanEqual=179, FOrdEqual=180, FOrdNotEqual=182,
    FOrdLessThan=184, FOrdGreaterThan=186, FOrdLessThanEqual=188, FOrdGreaterThanEqual=190, ShiftRightLogical=194,
    ShiftRightArithmetic=195, ShiftLeftLogical=196, BitwiseOr=197, BitwiseXor=198, BitwiseAnd=199, Not=200,
    BitCount=205, ControlBarrier=224, MemoryBarrier=225, AtomicIAdd=234, AtomicUMin=237, AtomicUMax=239, AtomicAnd=240,
    AtomicOr=241, Phi=245, LoopMerge=246, SelectionMerge=247, Label=248, Branch=249, BranchConditional=250, Switch=251,
    Return=253, Unreachable=255, ExecutionModeId=331, GroupNonUniformElect=333, GroupNonUniformBallot=339,
    CopyLogical=400, SpecConstantComposite=51, EmitMeshTasksEXT=5294,
)
OPN = {v: k for k, v in OP.items()}
DEC_SPEC_ID, DEC_ARRAY_STRIDE, DEC_MATRIX_STRIDE, DEC_BUILTIN, DEC_BINDING, DEC_SET, DEC_OFFSET = 1, 6, 7, 11, 33, 34, 35
DEC_ROW_MAJOR = 4
BUILTIN = {24: "NumWorkgroups", 25: "WorkgroupSize", 26: "WorkgroupId", 27: "LocalInvocationId",
           28: "GlobalInvocationId", 29: "LocalInvocationIndex", 36: "SubgroupSize", 40: "SubgroupId",
           41: "SubgroupLocalInvocationId", 38: "NumSubgroups"}
SC_UNIFORM_CONSTANT, SC_INPUT, SC_WORKGROUP, SC_PRIVATE, SC_FUNCTION, SC_PUSH, SC_STORAGE = 0, 1, 4, 6, 7, 9, 12
SC_TASK_PAYLOAD = 5402  # TaskPayloadWorkgroupEXT: memory shared by the workgroup, handed to the mesh shaders


class Module:
    def __init__(self, path, spec=None):
        raw = open(path, "rb").read()
        w = struct.unpack("<%dI" % (len(raw) // 4), raw)
        assert w[0] == 0x07230203, "not a SPIR-V module"
        self.types, self.consts, self.names, self.member_names = {}, {}, {}, {}
        self.dec, self.mdec = {}, {}
        self.vars, self.blocks, self.order = {}, {}, []
        self.local_size, self.local_size_id = None, None
        self.spec = dict(spec or {})
        insts, i = [], 5
        while i < len(w):
            wc, op = w[i] >> 16, w[i] & 0xFFFF
            insts.append((op, w[i + 1:i + wc]))
            i += wc
        for op, a in insts:  # decorations first: constants and layouts need them
            if op == OP["Decorate"]:
                self.dec.setdefault(a[0], {})[a[1]] = a[2:]
            elif op == OP["MemberDecorate"]:
                self.mdec.setdefault(a[0], {}).setdefault(a[1], {})[a[2]] = a[3:]
            elif op == OP["Name"]:
                self.names[a[0]] = b"".join(struct.pack("<I", x) for x in a[1:]).split(b"\0")[0].decode()
            elif op == OP["MemberName"]:
                self.member_names.setdefault(a[0], {})[a[1]] = b"".join(
                    struct.pack("<I", x) for x in a[2:]).split(b"\0")[0].decode()
        cur = None
        self.spec_ops = []
        for op, a in insts:
            n = OPN.get(op)
            if n == "ExecutionMode" and a[1] == 17:
                self.local_size = tuple(a[2:5])
            elif n == "ExecutionModeId" and a[1] == 38:
                self.local_size_id = tuple(a[2:5])
            elif n == "TypeVoid": self.types[a[0]] = ("void",)
            elif n == "TypeBool": self.types[a[0]] = ("bool",)
            elif n == "TypeInt": self.types[a[0]] = ("int", a[1], a[2])
            elif n == "TypeFloat": self.types[a[0]] = ("float", a[1])
            elif n == "TypeVector": self.types[a[0]] = ("vec", a[1], a[2])
            elif n == "TypeMatrix": self.types[a[0]] = ("mat", a[1], a[2])
            elif n == "TypeImage": self.types[a[0]] = ("image",) + tuple(a[1:])
            elif n == "TypeSampler": self.types[a[0]] = ("sampler",)
            elif n == "TypeSampledImage": self.types[a[0]] = ("sampledimage", a[1])
            elif n == "TypeArray": self.types[a[0]] = ("array", a[1], a[2])
            elif n == "TypeRuntimeArray": self.types[a[0]] = ("rtarray", a[1])
            elif n == "TypeStruct": self.types[a[0]] = ("struct", tuple(a[1:]))
            elif n == "TypePointer": self.types[a[0]] = ("ptr", a[1], a[2])
            elif n == "TypeFunction": self.types[a[0]] = ("fn",)
            elif n in ("ConstantTrue", "ConstantFalse"): self.consts[a[1]] = (a[0], n == "ConstantTrue")
            elif n in ("Constant", "SpecConstant"):
                val = a[2]
                if n == "SpecConstant":
                    sid = self.dec.get(a[1], {}).get(DEC_SPEC_ID)
                    key = self.names.get(a[1])
                    if sid is not None and sid[0] in self.spec: val = self.spec[sid[0]]
                    elif key in self.spec: val = self.spec[key]
                self.consts[a[1]] = (a[0], val)
            elif n in ("ConstantComposite", "SpecConstantComposite"): self.consts[a[1]] = (a[0], ("composite", a[2:]))
            elif n == "ConstantNull": self.consts[a[1]] = (a[0], ("null",))
            elif n == "Undef" and cur is None: self.consts[a[1]] = (a[0], ("null",))
            elif n == "SpecConstantOp": self.spec_ops.append(a)
            elif n == "Variable" and cur is None: self.vars[a[1]] = (a[0], a[2])
            elif n == "Label":
                cur = a[0]
                self.blocks[cur] = []
                self.order.append(cur)
            elif n == "FunctionEnd": cur = None
            elif cur is not None and op not in (8, 317): self.blocks[cur].append((op, a))  # OpLine / OpNoLine
        self.pos = {b: k for k, b in enumerate(self.order)}

    # -- layout helpers ---------------------------------------------------------------------------------------------
    def scalar_dtype(self, tid):
        t = self.types[tid]
        if t[0] == "float": return F32
        if t[0] == "int": return I32 if t[2] else U32
        if t[0] == "bool": return np.bool_
        raise TypeError(t)

    def size_of(self, tid, explicit=True):
        """Bytes of a value of type tid in memory (explicit = decorated buffer layout, else a natural layout)."""
        t = self.types[tid]
        if t[0] in ("float", "bool"): return 4
        if t[0] == "int": return t[1] // 8
        if t[0] == "vec": return self.size_of(t[1]) * t[2]
        if t[0] == "mat":
            return t[2] * (16 if explicit else self.size_of(t[1], False))
        if t[0] == "array":
            n = self.const_scalar(t[2])
            stride = self.dec.get(tid, {}).get(DEC_ARRAY_STRIDE)
            return n * (stride[0] if (explicit and stride) else self.size_of(t[1], explicit))
        if t[0] == "struct":
            if explicit and tid in self.mdec and all(DEC_OFFSET in self.mdec[tid].get(k, {}) for k in range(len(t[1]))):
                last = len(t[1]) - 1
                return self.mdec[tid][last][DEC_OFFSET][0] + self.size_of(t[1][last], True)
            return sum(self.size_of(m, False) for m in t[1])
        raise TypeError(t)

    def push_constant_layout(self):
        """{member name: (byte offset, byte size)} of the push-constant block, from the module's own decorations."""
        for vid, (ptid, sc) in self.vars.items():
            if sc == SC_PUSH:
                tid = self.types[ptid][2]
                return {self.member_names.get(tid, {}).get(k, str(k)): (self.mdec[tid][k][DEC_OFFSET][0], self.size_of(mt))
                        for k, mt in enumerate(self.types[tid][1])}
        return {}

    def const_scalar(self, cid):
        tid, v = self.consts[cid]
        return v


def fma32(a, b, c):
    """Correctly rounded fused multiply-add of binary32 arrays (through binary64 with a double-rounding repair)."""
    a64, b64, c64 = a.astype(np.float64), b.astype(np.float64), c.astype(np.float64)
    p = a64 * b64                      # exact: 24 x 24 bits
    s = p + c64                        # rounded to 53 bits
    bb = s - p                         # TwoSum: err = exact (p + c) - s
    err = (p - (s - bb)) + (c64 - bb)
    r = s.astype(F32)
    # double rounding matters only when s sits exactly half way between two binary32 values and err != 0
    bits = s.view(np.uint64) if isinstance(s, np.ndarray) else np.array(s).view(np.uint64)
    half = (bits & np.uint64(0x1FFFFFFF)) == np.uint64(0x10000000)
    fix = half & (err != 0) & np.isfinite(s) & (np.abs(s) >= 2.0 ** -126)
    if np.any(fix):
        lo = np.nextafter(r, F32(-np.inf))
        hi = np.nextafter(r, F32(np.inf))
        r64 = r.astype(np.float64)
        # the two candidates around s
        down = np.where(r64 > s, lo, r)
        up = np.where(r64 > s, r, hi)
        r = np.where(fix, np.where(err > 0, up, down), r)  # the exact sum lies on err's side of the half-way point
    return r.astype(F32)


class Ptr:
    __slots__ = ("mem", "off", "tid", "explicit", "per_lane")

    def __init__(self, mem, off, tid, explicit, per_lane):
        self.mem, self.off, self.tid, self.explicit, self.per_lane = mem, off, tid, explicit, per_lane


class Machine:
    """Runs entry point `main` of a Module for whole workgroups."""

    def __init__(self, module, buffers, push, log2, images=None, samplers=None, sample=None, subgroup=32,
                 fused_dot=False, unfused_fma=False):
        """fused_dot: evaluate Dot / MatrixTimesVector / Length as fma chains (a0*b0, then fma(a1, b1, acc), ...) —
        what AMD's and other drivers' compilers typically emit — instead of the canonical separately rounded sums.
        unfused_fma: evaluate GLSL.std.450 Fma as a rounded product followed by a rounded sum — what a driver MAY do
        with an Fma that is not decorated NoContraction (the binaries' are not).
        Only tools/spirv_sensitivity.py uses them, to measure how much of the output depends on those driver choices."""
        self.m, self.buffers, self.push = module, buffers, np.frombuffer(bytes(push), np.uint8).copy()
        self.log2, self.images, self.samplers, self.sample, self.S = log2, images or {}, samplers or {}, sample, subgroup
        self.fused_dot, self.unfused_fma = fused_dot, unfused_fma

    # -- values -----------------------------------------------------------------------------------------------------
    def const(self, cid, N):
        tid, v = self.m.consts[cid]
        return self.const_of(tid, v, N)

    def const_of(self, tid, v, N):
        t = self.m.types[tid]
        if t[0] == "bool": return np.full(N, bool(v) if not isinstance(v, tuple) else False)
        if t[0] == "int":
            x = 0 if isinstance(v, tuple) else v
            return np.full(N, x, U32).view(I32 if t[2] else U32).copy()
        if t[0] == "float":
            x = 0 if isinstance(v, tuple) else v
            return np.full(N, x, U32).view(F32).copy()
        if isinstance(v, tuple) and v[0] == "null":
            if t[0] == "vec": return np.stack([self.const_of(t[1], ("null",), N) for _ in range(t[2])], axis=1)
            if t[0] == "mat": return np.stack([self.const_of(t[1], ("null",), N) for _ in range(t[2])], axis=1)
            if t[0] == "array": return [self.const_of(t[1], ("null",), N) for _ in range(self.m.const_scalar(t[2]))]
            if t[0] == "struct": return [self.const_of(mt, ("null",), N) for mt in t[1]]
        if isinstance(v, tuple) and v[0] == "composite":
            parts = [self.val(c, N) for c in v[1]]
            if t[0] in ("vec", "mat"): return np.stack(parts, axis=1)
            return parts
        raise TypeError((t, v))

    def val(self, i, N=None):
        if i in self.vals: return self.vals[i]
        if i in self.m.consts:
            v = self.const(i, self.N)
            self.vals[i] = v
            return v
        raise KeyError("id %%%d (%s) has no value" % (i, self.m.names.get(i, "")))

    def setval(self, i, new, mask):
        old = self.vals.get(i)
        self.vals[i] = new if old is None else self.merge(old, new, mask)

    def merge(self, old, new, mask):
        if isinstance(new, list): return [self.merge(o, n, mask) for o, n in zip(old, new)]
        if isinstance(new, (Ptr, dict)) or new is None: return new
        m = mask.reshape((-1,) + (1,) * (new.ndim - 1))
        return np.where(m, new, old)

    # -- memory -----------------------------------------------------------------------------------------------------
    def load(self, p, mask, tid=None, off=None):
        m = self.m
        tid = p.tid if tid is None else tid
        off = p.off if off is None else off
        t = m.types[tid]
        if t[0] in ("float", "int", "bool"):
            o = np.where(mask, off, 0).astype(np.int64)
            nb = m.size_of(tid)
            if p.per_lane:
                rows = np.arange(self.N)
                b = [p.mem[rows, o + k].astype(np.uint64) for k in range(nb)]
            else:
                if np.any(o + nb > p.mem.size): raise IndexError("load past the end of a buffer")
                b = [p.mem[o + k].astype(np.uint64) for k in range(nb)]
            u = b[0]
            for k in range(1, nb): u = u | (b[k] << np.uint64(8 * k))
            if t[0] == "int" and t[2] and nb < 4:  # narrow signed integers live sign-extended in 32-bit lanes
                sign = np.uint64(1 << (8 * nb - 1))
                u = np.where(u & sign, u | np.uint64(0xFFFFFFFF & ~((1 << (8 * nb)) - 1)), u)
            u = u.astype(U32)
            if t[0] == "bool": return u != 0
            return u.view(m.scalar_dtype(tid)).copy()
        if t[0] == "vec":
            es = m.size_of(t[1])
            return np.stack([self.load(p, mask, t[1], off + es * k) for k in range(t[2])], axis=1)
        if t[0] == "mat":
            stride = 16 if p.explicit else m.size_of(t[1], False)
            return np.stack([self.load(p, mask, t[1], off + stride * c) for c in range(t[2])], axis=1)
        if t[0] == "array":
            n, stride = m.const_scalar(t[2]), self.stride(tid, p.explicit)
            return [self.load(p, mask, t[1], off + stride * k) for k in range(n)]
        if t[0] == "struct":
            return [self.load(p, mask, mt, off + self.member_offset(tid, k, p.explicit)) for k, mt in enumerate(t[1])]
        raise TypeError(t)

    def store(self, p, v, mask, tid=None, off=None):
        m = self.m
        tid = p.tid if tid is None else tid
        off = p.off if off is None else off
        t = m.types[tid]
        if t[0] in ("float", "int", "bool"):
            u = (v.astype(U32) if t[0] == "bool" else np.ascontiguousarray(v).view(U32))
            b = np.ascontiguousarray(u).view(np.uint8).reshape(-1, 4)
            nb = m.size_of(tid)
            for lane in np.flatnonzero(mask):  # lane order: the canonical order of conflicting stores
                o = int(off[lane])
                if p.per_lane: p.mem[lane, o:o + nb] = b[lane, :nb]
                else:
                    if o + nb > p.mem.size: raise IndexError("store past the end of a buffer")
                    p.mem[o:o + nb] = b[lane, :nb]
            return
        if t[0] == "vec":
            es = m.size_of(t[1])
            for k in range(t[2]): self.store(p, v[:, k], mask, t[1], off + es * k)
        elif t[0] == "mat":
            stride = 16 if p.explicit else m.size_of(t[1], False)
            for c in range(t[2]): self.store(p, v[:, c], mask, t[1], off + stride * c)
        elif t[0] == "array":
            stride = self.stride(tid, p.explicit)
            for k in range(m.const_scalar(t[2])): self.store(p, v[k], mask, t[1], off + stride * k)
        elif t[0] == "struct":
            for k, mt in enumerate(t[1]): self.store(p, v[k], mask, mt, off + self.member_offset(tid, k, p.explicit))
        else:
            raise TypeError(t)

    def stride(self, tid, explicit):
        d = self.m.dec.get(tid, {}).get(DEC_ARRAY_STRIDE)
        if explicit and d: return d[0]
        return self.m.size_of(self.m.types[tid][1], explicit)

    def member_offset(self, tid, k, explicit):
        md = self.m.mdec.get(tid, {}).get(k, {})
        if explicit and DEC_OFFSET in md: return md[DEC_OFFSET][0]
        return sum(self.m.size_of(mt, False) for mt in self.m.types[tid][1][:k])

    # -- arithmetic -------------------------------------------------------------------------------------------------
    def dot(self, a, b):
        acc = (a[:, 0] * b[:, 0]).astype(F32)
        for k in range(1, a.shape[1]):
            acc = fma32(a[:, k], b[:, k], acc) if self.fused_dot else (acc + a[:, k] * b[:, k]).astype(F32)
        return acc.astype(F32)

    def mat_vec(self, M, v):  # M[lane, col, row]
        rows = []
        for r in range(M.shape[2]):
            acc = (M[:, 0, r] * v[:, 0]).astype(F32)
            for c in range(1, M.shape[1]):
                acc = fma32(M[:, c, r], v[:, c], acc) if self.fused_dot else (acc + M[:, c, r] * v[:, c]).astype(F32)
            rows.append(acc.astype(F32))
        return np.stack(rows, axis=1)

    def ext(self, num, a, rt):
        x = a[0]
        if num == 4: return np.abs(x)
        if num == 8: return np.floor(x).astype(F32)
        if num == 9: return np.ceil(x).astype(F32)
        if num == 30: return self.log2(x)
        if num == 31: return np.sqrt(x).astype(F32)
        if num == 37: return np.where(a[1] < x, a[1], x)  # FMin: y < x ? y : x
        if num == 38: return np.minimum(x, a[1])           # UMin
        if num == 39: return np.minimum(x, a[1])
        if num == 40: return np.where(x < a[1], a[1], x)  # FMax: x < y ? y : x
        if num == 41: return np.maximum(x, a[1])
        if num == 42: return np.maximum(x, a[1])
        if num == 43:                                      # FClamp = min(max(x, lo), hi)
            t = np.where(x < a[1], a[1], x)
            return np.where(a[2] < t, a[2], t)
        if num == 44: return np.minimum(np.maximum(x, a[1]), a[2])
        if num == 45: return np.minimum(np.maximum(x, a[1]), a[2])
        if num == 50:
            if self.unfused_fma: return ((x * a[1]).astype(F32) + a[2]).astype(F32)
            return fma32(x, a[1], a[2]) if x.ndim == 1 else np.stack(
                [fma32(x[:, k], a[1][:, k], a[2][:, k]) for k in range(x.shape[1])], axis=1)
        if num == 66: return np.sqrt(self.dot(x, x)).astype(F32) if x.ndim == 2 else np.abs(x)
        if num == 67:
            d = (x - a[1]).astype(F32)
            return np.sqrt(self.dot(d, d)).astype(F32)
        raise NotImplementedError("GLSL.std.450 instruction %d" % num)

    # -- execution --------------------------------------------------------------------------------------------------
    def run(self, groups, local_size=None, batch=1):
        """Dispatch of groups = nx, (nx, ny) or (nx, ny, nz) workgroups.  `batch` workgroups (consecutive in x-major order) are
        interpreted side by side (their invocations are lanes of one SIMT group: same results, fewer numpy calls) —
        only for shaders without workgroup-shared memory or barriers; atomics still see the invocations in dispatch
        order (workgroup by workgroup, local index ascending)."""
        m = self.m
        if isinstance(groups, int): groups = (groups,)
        groups = tuple(groups) + (1,) * (3 - len(groups))
        if local_size is None:
            if m.local_size: local_size = m.local_size
            else: local_size = tuple(m.const_scalar(i) for i in m.local_size_id)
        if isinstance(local_size, int): local_size = (local_size,)
        local_size = tuple(local_size) + (1,) * (3 - len(local_size))
        total = groups[0] * groups[1] * groups[2]
        self.emitted = []  # task shaders: (group counts, payload) per workgroup, in dispatch order
        for g in range(0, total, batch):
            self.run_group(g, groups, local_size, min(batch, total - g))

    def run_group(self, g, groups, local_size, count=1):
        m = self.m
        lx, ly, lz = local_size
        L = lx * ly * lz
        N = L * count
        self.N, self.vals = N, {}
        lanes = np.arange(N, dtype=U32)
        local = lanes % U32(L)
        wg = U32(g) + lanes // U32(L)
        wgx, wgy, wgz = wg % U32(groups[0]), (wg // U32(groups[0])) % U32(groups[1]), wg // U32(groups[0] * groups[1])
        lix, liy, liz = local % U32(lx), (local // U32(lx)) % U32(ly), local // U32(lx * ly)
        z = np.zeros(N, U32)
        builtin = {"GlobalInvocationId": np.stack([wgx * U32(lx) + lix, wgy * U32(ly) + liy, wgz * U32(lz) + liz], axis=1),
                   "LocalInvocationId": np.stack([lix, liy, liz], axis=1), "LocalInvocationIndex": local,
                   "WorkgroupId": np.stack([wgx, wgy, wgz], axis=1),
                   "NumWorkgroups": np.stack([z + U32(groups[0]), z + U32(groups[1]), z + U32(groups[2])], axis=1),
                   "SubgroupSize": np.full(N, self.S, U32), "SubgroupLocalInvocationId": local % U32(self.S),
                   "SubgroupId": local // U32(self.S), "NumSubgroups": np.full(N, (L + self.S - 1) // self.S, U32)}
        # SpecConstantOps (only IAdd of a spec constant and a constant occurs)
        for a in m.spec_ops:
            rt, rid, opc = a[0], a[1], a[2]
            x, y = self.val(a[3]), self.val(a[4])
            assert OPN[opc] in ("IAdd", "ISub", "IMul", "UDiv"), OPN.get(opc)
            r = {"IAdd": x + y, "ISub": x - y, "IMul": x * y, "UDiv": x // np.maximum(y, 1)}[OPN[opc]]
            self.vals[rid] = r.astype(m.scalar_dtype(rt))
        for vid, (ptid, sc) in m.vars.items():
            pointee = m.types[ptid][2]
            if sc == SC_INPUT:
                bi = m.dec[vid][DEC_BUILTIN][0]
                v = builtin[BUILTIN[bi]]
                mem = np.zeros((N, 16), np.uint8)
                p = Ptr(mem, np.zeros(N, np.int64), pointee, False, True)
                self.store(p, v, np.ones(N, bool))
                self.vals[vid] = p
            elif sc == SC_PUSH:
                self.vals[vid] = Ptr(self.push, np.zeros(N, np.int64), pointee, True, False)
            elif sc == SC_STORAGE:
                self.vals[vid] = {"registry": "buffer", "tid": pointee}
            elif sc == SC_UNIFORM_CONSTANT:
                self.vals[vid] = {"registry": "handle", "tid": pointee}
            elif sc in (SC_WORKGROUP, SC_TASK_PAYLOAD):
                assert count == 1, "workgroup-shared memory: one workgroup per SIMT group"
                mem = np.zeros(max(m.size_of(pointee, False), 4), np.uint8)
                self.vals[vid] = Ptr(mem, np.zeros(N, np.int64), pointee, False, False)
                if sc == SC_TASK_PAYLOAD: self.payload_var = vid
            else:
                raise NotImplementedError("storage class %d" % sc)
        pcs = np.zeros(N, np.int64)          # position of every lane's current block; -1 = returned
        prev = np.full(N, -1, np.int64)      # label of the block a lane came from (Phi)
        with np.errstate(all="ignore"):
            while True:
                live = pcs >= 0
                if not live.any(): break
                b = int(pcs[live].min())     # structured control flow: the earliest block first = reconvergence
                mask = pcs == b
                self.exec_block(m.order[b], mask, pcs, prev)

    def exec_block(self, label, mask, pcs, prev):
        m, N = self.m, self.N
        insts = m.blocks[label]
        # Phis read their operands "at once"
        phis = [(a[0], a[1], a[2:]) for op, a in insts if op == OP["Phi"]]
        if phis:
            news = []
            for rt, rid, pairs in phis:
                acc = None
                for k in range(0, len(pairs), 2):
                    sel = mask & (prev == pairs[k + 1])
                    if not sel.any(): continue
                    v = self.val(pairs[k])
                    acc = v if acc is None else self.merge(acc, v, sel)
                news.append((rid, acc))
            for rid, acc in news:
                if acc is not None: self.setval(rid, acc, mask)
        for op, a in insts:
            n = OPN.get(op)
            if n is None: raise NotImplementedError("opcode %d" % op)
            if n in ("Phi", "LoopMerge", "SelectionMerge", "ControlBarrier", "MemoryBarrier"): continue
            if n == "Branch":
                prev[mask] = label
                pcs[mask] = m.pos[a[0]]
                return
            if n == "BranchConditional":
                c = self.val(a[0])
                prev[mask] = label
                pcs[mask & c] = m.pos[a[1]]
                pcs[mask & ~c] = m.pos[a[2]]
                return
            if n == "Switch":
                sel = self.val(a[0])
                prev[mask] = label
                tgt = np.full(N, m.pos[a[1]], np.int64)
                for k in range(2, len(a), 2):
                    tgt[sel.view(U32) == U32(a[k])] = m.pos[a[k + 1]]
                pcs[mask] = tgt[mask]
                return
            if n in ("Return", "Unreachable"):
                pcs[mask] = -1
                return
            if n == "EmitMeshTasksEXT":  # terminates the task shader: group counts + the payload as it stands
                first = int(np.flatnonzero(mask)[0])
                counts = tuple(int(self.val(a[k]).view(U32)[first]) for k in range(3))
                pay = self.load(self.vals[self.payload_var], mask) if getattr(self, "payload_var", None) else None
                self.emitted.append((counts, self.first_lane(pay, first)))
                pcs[mask] = -1
                return
            self.exec_inst(n, a, mask)
        raise RuntimeError("block %d has no terminator" % label)

    def first_lane(self, v, lane):
        if v is None: return None
        if isinstance(v, list): return [self.first_lane(x, lane) for x in v]
        return v[lane]

    def exec_inst(self, n, a, mask):
        m, N, V = self.m, self.N, self.val
        if n == "Variable":  # function storage: private memory per lane
            pointee = m.types[a[0]][2]
            if a[1] not in self.vals:
                mem = np.zeros((N, max(m.size_of(pointee, False), 4)), np.uint8)
                self.vals[a[1]] = Ptr(mem, np.zeros(N, np.int64), pointee, False, True)
            return
        if n == "Store":
            return self.store(V(a[0]), V(a[1]), mask)
        if n == "ImageWrite":
            img, coord, texel = V(a[0]), V(a[1]), V(a[2])
            for lane in np.flatnonzero(mask):
                if coord.shape[1] == 3: img["write3"](coord[lane], texel[lane])
                else: img["write"](int(coord[lane, 0]), int(coord[lane, 1]), texel[lane])
            return
        rt, rid, x = a[0], a[1], a[2:]
        t = m.types[rt]
        r = None
        if n in ("AccessChain", "InBoundsAccessChain"):
            base = V(x[0])
            idx = x[1:]
            if isinstance(base, dict):  # descriptor array: the first index picks the resource
                sel = V(idx[0])
                k = int(sel[np.flatnonzero(mask)[0]])
                assert (sel[mask] == sel[np.flatnonzero(mask)[0]]).all(), "non-uniform resource index"
                if base["registry"] == "buffer":
                    p = Ptr(self.buffers[k], np.zeros(N, np.int64), m.types[base["tid"]][1], True, False)
                    idx = idx[1:]
                else:
                    elem = m.types[base["tid"]]
                    kind = m.types[elem[1]][0]
                    r = (self.images if kind == "image" else self.samplers)[k]
                    self.setval(rid, r, mask)
                    return
            else:
                p = Ptr(base.mem, base.off.copy(), base.tid, base.explicit, base.per_lane)
            off, tid = p.off.astype(np.int64), p.tid
            for i in idx:
                tt = m.types[tid]
                if tt[0] == "struct":
                    k = m.const_scalar(i)
                    off = off + self.member_offset(tid, k, p.explicit)
                    tid = tt[1][k]
                elif tt[0] in ("array", "rtarray"):
                    iv = V(i).astype(np.int64) & 0xFFFFFFFF
                    off = off + iv * self.stride(tid, p.explicit)
                    tid = tt[1]
                elif tt[0] == "vec":
                    off = off + (V(i).astype(np.int64) & 0xFFFFFFFF) * m.size_of(tt[1])
                    tid = tt[1]
                elif tt[0] == "mat":
                    stride = 16 if p.explicit else m.size_of(tt[1], False)
                    off = off + (V(i).astype(np.int64) & 0xFFFFFFFF) * stride
                    tid = tt[1]
                else:
                    raise TypeError(tt)
            r = Ptr(p.mem, off, tid, p.explicit, p.per_lane)
        elif n == "Load":
            p = V(x[0])
            r = p if isinstance(p, dict) else self.load(p, mask)
        elif n in ("CopyObject", "CopyLogical"):
            r = V(x[0])
        elif n == "Undef":
            r = self.const_of(rt, ("null",), N)
        elif n == "CompositeConstruct":
            parts = [V(i) for i in x]
            if t[0] == "vec":
                cols = []
                for q in parts: cols += [q] if q.ndim == 1 else [q[:, k] for k in range(q.shape[1])]
                r = np.stack(cols, axis=1)
            elif t[0] == "mat": r = np.stack(parts, axis=1)
            else: r = parts
        elif n == "CompositeExtract":
            r = V(x[0])
            for k in x[1:]: r = r[k] if isinstance(r, list) else r[:, k]
        elif n == "CompositeInsert":
            obj, comp = V(x[0]), V(x[1])
            def ins(c, path):
                if isinstance(c, list):
                    c = list(c)
                    c[path[0]] = obj if len(path) == 1 else ins(c[path[0]], path[1:])
                    return c
                c = c.copy()
                if len(path) == 1: c[:, path[0]] = obj
                else: c[:, path[0], path[1]] = obj
                return c
            r = ins(comp, list(x[2:]))
        elif n == "VectorShuffle":
            v1, v2 = V(x[0]), V(x[1])
            cat = np.concatenate([v1, v2], axis=1)
            r = np.stack([cat[:, k] if k != 0xFFFFFFFF else cat[:, 0] * 0 for k in x[2:]], axis=1)
        elif n == "ExtInst":
            r = self.ext(x[1], [V(i) for i in x[2:]], rt)
        elif n in ("FAdd", "FSub", "FMul", "FDiv"):
            p, q = V(x[0]), V(x[1])
            r = {"FAdd": p + q, "FSub": p - q, "FMul": p * q, "FDiv": p / q}[n].astype(F32)
        elif n == "FNegate": r = (-V(x[0])).astype(F32)
        elif n == "VectorTimesScalar": r = (V(x[0]) * V(x[1])[:, None]).astype(F32)
        elif n == "Dot": r = self.dot(V(x[0]), V(x[1]))
        elif n == "MatrixTimesVector": r = self.mat_vec(V(x[0]), V(x[1]))
        elif n == "MatrixTimesMatrix":
            A, B = V(x[0]), V(x[1])
            r = np.stack([self.mat_vec(A, B[:, c]) for c in range(B.shape[1])], axis=1)
        elif n in ("IAdd", "ISub", "IMul", "UDiv", "SDiv", "UMod", "BitwiseAnd", "BitwiseOr", "BitwiseXor",
                   "ShiftLeftLogical", "ShiftRightLogical", "ShiftRightArithmetic"):
            dt = m.scalar_dtype(rt if t[0] == "int" else t[1])
            p, q = V(x[0]).view(U32).astype(np.uint64), V(x[1]).view(U32).astype(np.uint64)
            if n == "IAdd": u = p + q
            elif n == "ISub": u = p - q
            elif n == "IMul": u = p * q
            elif n == "UDiv": u = p // np.maximum(q, 1)
            elif n == "UMod": u = p % np.maximum(q, 1)
            elif n == "SDiv":
                ps, qs = V(x[0]).view(U32).view(I32).astype(np.int64), V(x[1]).view(U32).view(I32).astype(np.int64)
                u = (np.trunc(ps / np.where(qs == 0, 1, qs))).astype(np.int64).astype(np.uint64)
            elif n == "BitwiseAnd": u = p & q
            elif n == "BitwiseOr": u = p | q
            elif n == "BitwiseXor": u = p ^ q
            elif n == "ShiftLeftLogical": u = np.where(q < 32, p << np.minimum(q, 31), 0)
            elif n == "ShiftRightLogical": u = np.where(q < 32, p >> np.minimum(q, 31), 0)
            else:
                ps = V(x[0]).view(U32).view(I32).astype(np.int64)
                u = (ps >> np.minimum(q, 31).astype(np.int64)).astype(np.uint64)
            r = (u & np.uint64(0xFFFFFFFF)).astype(U32).view(dt)
        elif n == "SNegate":
            r = (-(V(x[0]).view(U32).view(I32).astype(np.int64))).astype(np.uint64).astype(U32).view(m.scalar_dtype(rt))
        elif n == "Not":
            r = (~V(x[0]).view(U32)).view(m.scalar_dtype(rt if t[0] == "int" else t[1]))
        elif n == "BitCount":
            u = V(x[0]).view(U32).astype(np.uint64)
            c = np.zeros_like(u)
            for k in range(32): c += (u >> np.uint64(k)) & np.uint64(1)
            r = c.astype(U32).view(m.scalar_dtype(rt if t[0] == "int" else t[1]))
        elif n in ("LogicalAnd", "LogicalOr"):
            r = (V(x[0]) & V(x[1])) if n == "LogicalAnd" else (V(x[0]) | V(x[1]))
        elif n == "LogicalNot": r = ~V(x[0])
        elif n == "Select":
            c, p, q = V(x[0]), V(x[1]), V(x[2])
            if isinstance(p, list): r = [np.where(c.reshape((-1,) + (1,) * (pp.ndim - 1)), pp, qq) for pp, qq in zip(p, q)]
            else: r = np.where(c.reshape((-1,) + (1,) * (p.ndim - c.ndim)) if c.ndim < p.ndim else c, p, q)
        elif n in ("IEqual", "INotEqual"):
            e = V(x[0]).view(U32) == V(x[1]).view(U32)
            r = e if n == "IEqual" else ~e
        elif n in ("UGreaterThan", "UGreaterThanEqual", "ULessThan", "ULessThanEqual"):
            p, q = V(x[0]).view(U32), V(x[1]).view(U32)
            r = {"UGreaterThan": p > q, "UGreaterThanEqual": p >= q, "ULessThan": p < q, "ULessThanEqual": p <= q}[n]
        elif n in ("SGreaterThan", "SGreaterThanEqual", "SLessThan", "SLessThanEqual"):
            p, q = V(x[0]).view(U32).view(I32), V(x[1]).view(U32).view(I32)
            r = {"SGreaterThan": p > q, "SGreaterThanEqual": p >= q, "SLessThan": p < q, "SLessThanEqual": p <= q}[n]
        elif n in ("FOrdEqual", "FOrdNotEqual", "FOrdLessThan", "FOrdGreaterThan", "FOrdLessThanEqual",
                   "FOrdGreaterThanEqual"):
            p, q = V(x[0]), V(x[1])
            r = {"FOrdEqual": p == q, "FOrdNotEqual": (p != q) & ~np.isnan(p) & ~np.isnan(q), "FOrdLessThan": p < q,
                 "FOrdGreaterThan": p > q, "FOrdLessThanEqual": p <= q, "FOrdGreaterThanEqual": p >= q}[n]
        elif n == "ConvertSToF": r = V(x[0]).view(U32).view(I32).astype(F32)
        elif n == "ConvertUToF": r = V(x[0]).view(U32).astype(F32)
        elif n == "ConvertFToU":
            f = V(x[0])
            r = np.where(np.isnan(f) | (f <= -1), 0, np.minimum(np.trunc(f).astype(np.float64), 4294967295.0)).astype(np.uint64).astype(U32)
        elif n == "ConvertFToS":
            f = V(x[0])
            r = np.clip(np.where(np.isnan(f), 0, np.trunc(f).astype(np.float64)), -2147483648.0, 2147483647.0).astype(np.int64).astype(I32)
        elif n in ("UConvert", "SConvert"):  # 8/16-bit storage types are held in 32-bit lanes already
            r = V(x[0])
            if r.dtype not in (U32, I32): r = r.astype(I32 if n == "SConvert" else U32)
            r = r.view(U32).view(m.scalar_dtype(rt if t[0] == "int" else t[1]))
        elif n == "Bitcast":
            v = V(x[0])
            dt = m.scalar_dtype(rt if t[0] in ("int", "float") else t[1])
            r = np.ascontiguousarray(v).view(U32).view(dt) if v.dtype != np.bool_ else v
        elif n in ("AtomicIAdd", "AtomicUMax", "AtomicUMin", "AtomicOr", "AtomicAnd"):
            p, v = V(x[0]), V(x[3]).view(U32)
            old = np.zeros(N, U32)
            for lane in np.flatnonzero(mask):
                o = int(p.off[lane])
                mem = p.mem[lane] if p.per_lane else p.mem
                cur = int(np.frombuffer(mem[o:o + 4].tobytes(), U32)[0])
                old[lane] = cur
                y = int(v[lane])
                new = {"AtomicIAdd": (cur + y) & 0xFFFFFFFF, "AtomicUMax": max(cur, y), "AtomicUMin": min(cur, y),
                       "AtomicOr": cur | y, "AtomicAnd": cur & y}[n]
                mem[o:o + 4] = np.frombuffer(struct.pack("<I", new), np.uint8)
            r = old.view(m.scalar_dtype(rt))
        elif n == "GroupNonUniformElect":
            r = np.zeros(N, bool)
            for s0 in range(0, N, self.S):
                act = np.flatnonzero(mask[s0:s0 + self.S])
                if act.size: r[s0 + act[0]] = True
        elif n == "GroupNonUniformBallot":
            pred = V(x[1]) & mask
            r = np.zeros((N, 4), U32)
            for s0 in range(0, N, self.S):
                words = [0, 0, 0, 0]
                for k in np.flatnonzero(pred[s0:s0 + self.S]): words[k // 32] |= 1 << (k % 32)
                r[s0:s0 + self.S] = np.array(words, U32)
        elif n == "SampledImage":
            r = {"image": V(x[0]), "sampler": V(x[1])}
        elif n == "Image":
            r = V(x[0])["image"]
        elif n == "ImageQuerySizeLod":
            img, lod = V(x[0]), V(x[1]).view(U32)
            r = np.zeros((N, 2), I32)
            for lane in np.flatnonzero(mask):
                w, h = img["size"](int(lod[lane]))
                r[lane] = (w, h)
        elif n == "ImageQuerySize":
            img = V(x[0])
            w, h = img["size"](0)
            r = np.tile(np.array([w, h], I32), (N, 1))
        elif n == "ImageSampleExplicitLod":
            si, coord, lod = V(x[0]), V(x[1]), V(x[3])
            assert x[2] == 2, "only the Lod image operand is implemented"
            out = np.zeros((N, 4), F32)
            for lane in np.flatnonzero(mask):
                out[lane] = self.sample(si["image"], si["sampler"], F32(coord[lane, 0]), F32(coord[lane, 1]), F32(lod[lane]))
            r = out
        elif n == "ImageFetch":
            img, coord = V(x[0]), V(x[1])
            out = np.zeros((N, 4), F32)
            act = np.flatnonzero(mask)
            if len(x) > 2 and x[2] & 0x40:  # Sample image operand: a multisampled image
                smp = V(x[3]).view(U32)
                out[act] = img["fetch_ms"](coord[act, 0].astype(np.int64), coord[act, 1].astype(np.int64),
                                           smp[act].astype(np.int64))
            else:
                out[act] = img["fetch"](coord[act, 0].astype(np.int64), coord[act, 1].astype(np.int64))
            r = out
        else:
            raise NotImplementedError(n)
        self.setval(rid, r, mask)

"""Pins the gfx950 register layouts the kernels are written against (guide cdna_hip_programming §3):
the same one-wave probe (csrc/debug_probe.hip) runs through the host emulator — which encodes OUR
reading of the MFMA operand/accumulator maps and of ds_read_b64_tr_b16 — and on the GPU; outputs
must be identical (inputs are small integers, so MFMA sums are exact in fp32)."""
import ctypes as C

import numpy as np
import pytest
import torch


def _run_probe(library, device, which, in_bytes: bytes, out_nbytes: int) -> bytes:
    buf = torch.frombuffer(bytearray(in_bytes), dtype=torch.uint8).to(device)
    out = torch.zeros(out_nbytes, dtype=torch.uint8, device=device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if device != "cpu" else None
    library.call("vq_debug_probe", which, C.c_void_p(buf.data_ptr()), C.c_void_p(out.data_ptr()), stream)
    if device != "cpu":
        torch.cuda.synchronize()
    return bytes(out.cpu().numpy().tobytes())


def _bf16_bits(x: np.ndarray) -> np.ndarray:
    return (x.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)


def _f16_bits(x: np.ndarray) -> np.ndarray:
    return x.astype(np.float16).view(np.uint16)


def _mfma_input(seed, bits=_bf16_bits):
    rng = np.random.default_rng(seed)
    a = rng.integers(-4, 5, size=(64, 8))
    b = rng.integers(-4, 5, size=(64, 8))
    return np.concatenate([bits(a).reshape(-1), bits(b).reshape(-1)]).tobytes()


def _tr_inputs():
    img = np.arange(1024, dtype=np.int16)
    lane = np.arange(64)
    linear = (lane * 4).astype(np.int32)
    gg, tl = lane >> 4, lane & 15
    rstr = 48
    wgrad = ((8 * (gg >> 1) + (tl >> 2)) * rstr + (gg & 1) * 16 + (tl & 3) * 4).astype(np.int32)
    return [img.tobytes() + off.tobytes() for off in (linear, wgrad)]


def test_emulator_mfma_matches_matrix_product(emu_library):
    """The emulator's 32x32x16 map really is D = A @ B under the documented lane layout."""
    rng = np.random.default_rng(0)
    A = rng.integers(-4, 5, size=(32, 16)); B = rng.integers(-4, 5, size=(16, 32))
    a = np.zeros((64, 8)); b = np.zeros((64, 8))
    for l in range(64):
        for t in range(8):
            a[l, t] = A[l & 31, 8 * (l >> 5) + t]
            b[l, t] = B[8 * (l >> 5) + t, l & 31]
    inp = np.concatenate([_bf16_bits(a).reshape(-1), _bf16_bits(b).reshape(-1)]).tobytes()
    out = np.frombuffer(_run_probe(emu_library, "cpu", 0, inp, 64 * 16 * 4), dtype=np.float32).reshape(64, 16)
    D = np.zeros((32, 32))
    for l in range(64):
        for r in range(16):
            D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = out[l, r]
    assert np.array_equal(D, A @ B)


def test_emulated_f16_mfma_equals_the_bf16_one_on_small_integers(emu_library):
    """Same shape, same lane maps, only the operand decoding differs (guide §3: layouts are dtype-independent)."""
    for seed in range(2):
        a = _run_probe(emu_library, "cpu", 0, _mfma_input(seed), 64 * 16 * 4)
        b = _run_probe(emu_library, "cpu", 3, _mfma_input(seed, _f16_bits), 64 * 16 * 4)
        assert a == b


@pytest.mark.gpu
@pytest.mark.parametrize("which,out_bytes", [(0, 64 * 16 * 4), (1, 64 * 4 * 4), (3, 64 * 16 * 4)])
def test_mfma_layout_matches_silicon(emu_library, hip_library, which, out_bytes):
    for seed in range(3):
        inp = _mfma_input(seed, _f16_bits if which == 3 else _bf16_bits)
        want = _run_probe(emu_library, "cpu", which, inp, out_bytes)
        got = _run_probe(hip_library, "cuda:0", which, inp, out_bytes)
        assert got == want, f"MFMA probe {which}: silicon layout differs from the emulated reading"


@pytest.mark.gpu
def test_lds_transpose_read_matches_silicon(emu_library, hip_library):
    for inp in _tr_inputs():
        want = np.frombuffer(_run_probe(emu_library, "cpu", 2, inp, 64 * 4 * 2), dtype=np.int16)
        got = np.frombuffer(_run_probe(hip_library, "cuda:0", 2, inp, 64 * 4 * 2), dtype=np.int16)
        assert np.array_equal(got, want), f"ds_read_b64_tr_b16: got {got.reshape(64, 4)[:20].tolist()} want {want.reshape(64, 4)[:20].tolist()}"


def _swap_probe(library, device):
    a = np.arange(64, dtype=np.uint32) + 1000
    b = np.arange(64, dtype=np.uint32) + 2000
    out = np.frombuffer(_run_probe(library, device, 4, a.tobytes() + b.tobytes(), 128 * 4), dtype=np.uint32)
    return a, b, out[:64], out[64:]


def test_emulated_permlane32_swap_is_what_the_epilogue_assumes(emu_library):
    """vq_swap32(a, b): lower lanes keep their a and receive the a of lane + 32 in b; upper lanes receive the b of lane - 32 in
    a and keep their b — what lets a lane assemble 8 consecutive output channels from the two halves of an MFMA accumulator."""
    a, b, na, nb = _swap_probe(emu_library, "cpu")
    assert np.array_equal(na[:32], a[:32]) and np.array_equal(nb[:32], a[32:])
    assert np.array_equal(na[32:], b[:32]) and np.array_equal(nb[32:], b[32:])


@pytest.mark.gpu
def test_permlane32_swap_matches_silicon(emu_library, hip_library):
    want = _swap_probe(emu_library, "cpu")
    got = _swap_probe(hip_library, "cuda:0")
    assert all(np.array_equal(g, w) for g, w in zip(got, want)), "v_permlane32_swap_b32 differs from the emulated reading"


def _f32_mfma_input(seed):
    """Full-mantissa operands: the result bits depend on the summation order (k = 0 before k = 1 onto the accumulator) and on the
    fusing (one rounding per step), so a bitwise match pins both."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(64).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    c = (rng.standard_normal((64, 16)) * (1.0 if seed % 2 else 1e-3)).astype(np.float32)
    return a.tobytes() + b.tobytes() + c.tobytes()


def test_emulated_f32_mfma_is_an_fmaf_chain(emu_library):
    """D[i][j] = fmaf(A[i][1], B[1][j], fmaf(A[i][0], B[0][j], C[i][j])) under the documented lane layout — what makes the MFMA-based
    nearest-code search (csrc/optim_vq.hip) reproduce oracle/vq_oracle.c's `dot = fmaf(z_k, e_k, dot)` chain bit for bit."""
    import math
    inp = _f32_mfma_input(1)
    out = np.frombuffer(_run_probe(emu_library, "cpu", 5, inp, 64 * 16 * 4), dtype=np.float32).reshape(64, 16)
    a = np.frombuffer(inp[:256], dtype=np.float32); b = np.frombuffer(inp[256:512], dtype=np.float32)
    c = np.frombuffer(inp[512:], dtype=np.float32).reshape(64, 16)
    fma = lambda x, y, z: np.float32(np.float64(x) * np.float64(y) + np.float64(z))     # exact product (48 bits) + one rounding: fmaf
    for l in (0, 17, 40, 63):
        for r in range(16):
            i, j = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31
            want = fma(a[i + 32], b[j + 32], fma(a[i], b[j], c[l, r]))
            assert out[l, r] == want, (l, r)


@pytest.mark.gpu
def test_f32_mfma_matches_silicon_bit_for_bit(emu_library, hip_library):
    for seed in range(4):
        inp = _f32_mfma_input(seed)
        want = _run_probe(emu_library, "cpu", 5, inp, 64 * 16 * 4)
        got = _run_probe(hip_library, "cuda:0", 5, inp, 64 * 16 * 4)
        assert got == want, "v_mfma_f32_32x32x2_f32: silicon differs from the emulated fmaf chain (layout, order or fusing)"

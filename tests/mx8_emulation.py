"""Torch emulation of the MX-fp8 operand format of the C4 GEMMs (OCP Microscaling v1.0: e4m3 elements, one E8M0 scale
per 32 K-elements, X = 2^(floor(log2 amax) - 8), saturating conversion).  Used by the CPU and GPU tests as the
independent statement of what tld_quant.hip / tld_gemm.hip (F8) must compute.  Not in the reference."""
import torch


def mx8_quantize(x: torch.Tensor):
    """x [R, K] float32 (K % 32 == 0) -> (e4m3 bytes [R, K] uint8, E8M0 bytes [R, K/32] uint8)."""
    R, K = x.shape
    xb = x.float().view(R, K // 32, 32)
    amax = xb.abs().amax(-1)
    e = torch.frexp(amax)[1] - 1                                  # floor(log2 amax) for amax > 0
    e8 = torch.where(amax > 0, (e + 127 - 8).clamp(min=0), torch.zeros_like(e))
    scale = torch.pow(2.0, (e8 - 127).double()).float()
    q = (xb / scale[..., None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    return q.view(R, K).view(torch.uint8), e8.to(torch.uint8)


def mx8_dequantize(q_u8: torch.Tensor, e8: torch.Tensor) -> torch.Tensor:
    R, K = q_u8.shape
    v = q_u8.view(torch.float8_e4m3fn).float().view(R, K // 32, 32)
    return (v.double() * torch.pow(2.0, (e8.double() - 127))[..., None]).view(R, K)


def scales_to_gemm_layout(e8: torch.Tensor) -> torch.Tensor:
    """[R, K/32] -> [K/128, R, 4]: the layout the GEMM's tile DMA reads (one contiguous KiB per 256 rows and K-step)."""
    R, nb = e8.shape
    return e8.view(R, nb // 4, 4).permute(1, 0, 2).contiguous()

#!/usr/bin/env python3
"""One-off assembler used in round 5 to cut DESIGN.md down: keeps the still-current sections of the round-4 text (profiles/notes/DESIGN_rounds_1-4.md)
by line range and splices the rewritten ones (tools/dbg/design_r5_*.md) between them."""
import os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
old = open(os.path.join(R, "profiles/notes/DESIGN_rounds_1-4.md")).read().split("\n")
L = lambda a, b: "\n".join(old[a - 1:b])
new = lambda n: open(os.path.join(R, "tools/dbg", f"design_r5_{n}.md")).read().rstrip("\n")
parts = [
    new("head"),            # title, section 0
    L(24, 104),             # 1. path, boundary, restructurings
    new("oracle"),          # 2.
    L(140, 161),            # 3. data layout
    new("kernels_intro"),   # 4. table of kernels (round-5 numbers)
    L(204, 253),            # 4.1 GEMM design bullets
    L(266, 288),            # fused depthwise epilogue (shipped form)
    L(289, 313),            # ring K loop
    new("gemm_tail"),       # pointer to the experiment tables
    new("attention"),       # 4.2 condensed
    L(452, 490),            # 4.3 row kernels
    L(491, 518),            # 4.4 fp8
    L(519, 545),            # 4.5 512 / 1024 px
    L(546, 574),            # 4.6 fused QKV + attention
    new("lowlat"),          # 4.7 low-latency class
    new("measurement"),     # 5.
    L(669, 690),            # 6. multi-GPU
    new("adjacent"),        # 7.
    new("limits"),          # 8.
    new("where"),           # 9.
]
doc = "\n\n".join(parts) + "\n"
doc = doc.replace("* **Second form of that epilogue (`EPI_UP_DWCONV2`, shipped; `TLD_UPDW_V2=0` selects the first).**  The K loop runs in",
                  "* **Fused depthwise 3×3 + GELU epilogue (`EPI_UP_DWCONV2`; the up-projection at 256 px).**  The tile's 256 rows are one 16 × 16-token image, so the MLP's depthwise conv is\n"
                  "  tile-local: `bf16(acc + bias)` goes to LDS as an image and the conv + GELU run from it — the pre-conv hidden tensor (a 200 MB write + read per layer) never reaches HBM.\n"
                  "  (\"Form 1\" below is the first version of this epilogue — token-major fp32 window, packed FMAs — retired in round 4.)  The K loop runs in")
doc = doc.replace("form 2 is kept because it leaves more registers and instruction slots for what comes next (§9).", "form 2 is the one kept.")
open(os.path.join(R, "DESIGN.md"), "w").write(doc)
print("DESIGN.md", sum(p.count("\n") + 1 for p in parts), "lines")

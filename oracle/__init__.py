"""CPU restatement of the reference's DiffSim scoring path -- TEST INFRASTRUCTURE, never imported by the product.

Pinning status (see cpu_ref.py's section headers and DESIGN.md section 2): everything whose arithmetic is the reference's own
Python is pinned by golden vectors generated from the reference itself (tests/golden/make_golden*.py, G1-G10).  The leaves that
live in un-vendored third-party packages (diffusers 0.29.2, timm 1.0.12: ResnetBlock2D, GroupNorm / LayerNorm eps, GEGLU,
samplers, time embeddings, PNDM / Euler tables, AutoencoderKL) are restated from their published semantics and stay
"parity unpinned" HERE: neither package nor a checkpoint exists in these containers.  tools/pin_with_real_diffusers.py drives
the real diffusers U-Net / VAE / scheduler and this oracle from one SD1.5 checkpoint on the repo's synthetic inputs and
compares q / k / v at every tap, the scores, the VAE moments and the scheduler table; it has to be run elsewhere (it needs
diffusers and the checkpoint) and has NOT been run by this build.
"""

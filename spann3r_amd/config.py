"""Model geometry of the Spann3R hot path.

The reference hard-wires this geometry through the DUSt3R checkpoint's constructor
string (/root/reference/dust3r/model.py:36-47) and through literal
1024/768 sizes in /root/reference/spann3r/model.py:225-261.  Only the depths are free.
"""
from dataclasses import dataclass, field, asdict
import re


@dataclass(frozen=True)
class Spann3RConfig:
    enc_dim: int = 1024      # CroCo ViT-L width (memory width too)
    enc_depth: int = 24
    enc_heads: int = 16
    dec_dim: int = 768       # ViT-B decoder width
    dec_depth: int = 12
    dec_heads: int = 12
    val_depth: int = 6       # value encoder: 6 x Block(1024, 16 heads)  (spann3r/model.py:228-235)
    patch: int = 16
    mlp_ratio: int = 4
    rope_base: float = 100.0  # pos_embed='RoPE100' (croco/models/croco.py:57-62)
    dpt_feat: int = 256      # dust3r/heads/dpt_head.py:101
    dpt_last: int = 128      # feature_dim // 2
    key_dim: int = 1792      # enc_dim + dec_dim (spann3r/model.py:250)
    mem_pos_enc: bool = False  # Spann3R(mem_pos_enc=True): RoPE in the value-encoder blocks (spann3r/model.py:232-234)
    use_feat: bool = False     # Spann3R(use_feat=True): the value encoder runs on dec1[-1] (768 wide, 16 heads of 48; :225,312-314)

    @property
    def head_dim(self):
        return self.enc_dim // self.enc_heads

    @property
    def val_dim(self):
        """width of the value encoder (spann3r/model.py:225: 768 if use_feat else 1024); its head count is always 16 (:228)"""
        return self.dec_dim if self.use_feat else self.enc_dim

    @property
    def hooks(self):
        # dust3r/heads/dpt_head.py:110: [0, l2*2//4, l2*3//4, l2]
        l2 = self.dec_depth
        return (0, l2 * 2 // 4, l2 * 3 // 4, l2)

    def as_dict(self):
        return asdict(self)

    @staticmethod
    def from_ctor_string(s: str) -> "Spann3RConfig":
        """Parse the constructor string a DUSt3R checkpoint carries in ckpt['args'].model
        (dust3r/model.py:36-47 eval()s it; we only read the integers we need)."""
        def grab(name, default):
            m = re.search(name + r"\s*=\s*(\d+)", s)
            return int(m.group(1)) if m else default
        cfg = Spann3RConfig(
            enc_dim=grab("enc_embed_dim", 1024), enc_depth=grab("enc_depth", 24),
            enc_heads=grab("enc_num_heads", 16), dec_dim=grab("dec_embed_dim", 768),
            dec_depth=grab("dec_depth", 12), dec_heads=grab("dec_num_heads", 12))
        if cfg.enc_dim != 1024 or cfg.dec_dim != 768:
            raise ValueError("Spann3R hard-wires 1024/768 widths (spann3r/model.py:245-250)")
        if cfg.dec_depth <= 9:
            raise ValueError("DPT head needs dec_depth > 9 (dust3r/heads/dpt_head.py:100)")
        return cfg

    def ctor_string(self, patch_embed_cls="ManyAR_PatchEmbed") -> str:
        """The string a reference-format checkpoint stores (used by the golden generator)."""
        return ("AsymmetricCroCo3DStereo(pos_embed='RoPE100', patch_embed_cls='%s', img_size=(512, 512), "
                "head_type='dpt', output_mode='pts3d', depth_mode=('exp', -inf, inf), conf_mode=('exp', 1, inf), "
                "enc_embed_dim=%d, enc_depth=%d, enc_num_heads=%d, dec_embed_dim=%d, dec_depth=%d, dec_num_heads=%d)"
                % (patch_embed_cls, self.enc_dim, self.enc_depth, self.enc_heads,
                   self.dec_dim, self.dec_depth, self.dec_heads))


FULL = Spann3RConfig()
TINY = Spann3RConfig(enc_depth=2, dec_depth=10)

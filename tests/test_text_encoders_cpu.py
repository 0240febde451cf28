"""Row a17 of SURVEY.md section 8 (`encode_prompts_flux`, toolkit/train_tools.py:510-574; FLUX plug-in hook `get_prompt_embeds`,
extensions_built_in/diffusion_models/flux_kontext/flux_kontext.py:354-367): a LIBRARY path — `transformers` text encoders on PyTorch, once per
caption, off the per-step path.  tests/golden/text_encoders_flux.safetensors = the reference's own function executed on tiny random CLIP / T5
encoders with a deterministic stand-in tokenizer (tests/golden/make_golden.py golden_text_encoders)."""
import hashlib
import os

import pytest
import torch
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd import plugin

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text_encoders_flux.safetensors")
PROMPTS = ["a photo of a red fox in the snow", "", "two words"]


class _TokOut(dict):
    def __init__(self, input_ids, attention_mask):
        super().__init__(input_ids=input_ids, attention_mask=attention_mask)
        self.input_ids, self.attention_mask = input_ids, attention_mask


class HashTokenizer:  # twin of make_golden.HashTokenizer
    def __init__(self, vocab_size, model_max_length, pad_id=0, eos_id=1):
        self.vocab_size, self.model_max_length, self.pad_id, self.eos_id = vocab_size, model_max_length, pad_id, eos_id

    def __call__(self, prompts, padding=None, max_length=None, truncation=True, return_tensors="pt", **kw):
        ids, mask = [], []
        for p in prompts:
            toks = [2 + int(hashlib.sha256(w.encode()).hexdigest(), 16) % (self.vocab_size - 2) for w in p.split()][: max_length - 1] + [self.eos_id]
            mask.append([1] * len(toks) + [0] * (max_length - len(toks)))
            ids.append(toks + [self.pad_id] * (max_length - len(toks)))
        return _TokOut(torch.tensor(ids), torch.tensor(mask))


def _encoders(seed=31):
    from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel

    torch.manual_seed(seed)
    clip = CLIPTextModel(CLIPTextConfig(vocab_size=99, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                        max_position_embeddings=16, eos_token_id=1, pad_token_id=0, bos_token_id=2)).eval()
    t5 = T5EncoderModel(T5Config(vocab_size=101, d_model=24, d_kv=8, d_ff=48, num_layers=2, num_heads=3, is_encoder_decoder=False, use_cache=False)).eval()
    return [HashTokenizer(99, 16), HashTokenizer(101, 512)], [clip, t5]


def test_encode_prompts_flux_equals_the_reference_function():
    gold = load_file(GOLD)
    toks, tes = _encoders()
    with torch.no_grad():
        for tag, kw in (("plain", {}), ("masked", {"attn_mask": True}), ("len64", {"max_length": 64})):
            emb, pooled = plugin.encode_prompts_flux(toks, tes, list(PROMPTS), **kw)
            assert torch.equal(emb, gold[f"{tag}/embeds"]) and torch.equal(pooled, gold[f"{tag}/pooled"]), tag
    assert not torch.equal(gold["plain/embeds"], gold["masked/embeds"])  # attention masking zeroes the padded positions
    assert gold["plain/embeds"].shape == (3, 512, 24) and gold["len64/embeds"].shape == (3, 64, 24)


def test_flux_plugin_get_prompt_embeds_hook():
    gold = load_file(GOLD)
    plug = plugin.Flux1MI355Model("cpu", dtype=torch.float32)
    plug.tokenizer, plug.text_encoder = _encoders()
    pe = plug.get_prompt_embeds(list(PROMPTS))
    assert torch.equal(pe.text_embeds, gold["plain/embeds"]) and torch.equal(pe.pooled_embeds, gold["plain/pooled"])
    one = plug.get_prompt_embeds(PROMPTS[0])  # a single string, like BaseModel.encode_prompt passes after wrapping (base_model.py:1108-1131)
    assert torch.equal(one.text_embeds, gold["plain/embeds"][:1]) and tuple(one.pooled_embeds.shape) == (1, 32)
    c = pe.clone().to(torch.float64).detach()
    assert c.text_embeds.dtype == torch.float64 and c.pooled_embeds.dtype == torch.float64
    # the embeddings feed the native step as they are: (text, pooled) through get_noise_prediction's _embeds
    assert plugin._embeds(pe)[0] is pe.text_embeds and plugin._embeds(pe)[1] is pe.pooled_embeds


def test_without_text_encoder_folders_the_hook_says_what_to_do(tmp_path):
    plug = plugin.Flux1MI355Model("cpu", __import__("types").SimpleNamespace(name_or_path=str(tmp_path), extras_name_or_path=None), dtype=torch.float32)
    with pytest.raises(FileNotFoundError, match="cache_text_embeddings"):
        plug.get_prompt_embeds("a photo")
    for other in (plugin.Wan21MI355Model, plugin.StableDiffusionMI355Model):
        with pytest.raises(FileNotFoundError, match="cache_text_embeddings"):
            other("cpu", __import__("types").SimpleNamespace(name_or_path=str(tmp_path), extras_name_or_path=None), dtype=torch.float32).get_prompt_embeds("a photo")


def test_wan_prompt_encoding_cuts_at_the_sequence_length_and_zero_pads():
    """Wan2.1: PARITY UNPINNED (diffusers' WanPipeline.encode_prompt is not vendored) — the published behaviour is checked: text cleaned, UMT5 last
    hidden state kept up to each prompt's own token count, zeros from there to max_sequence_length"""
    from transformers import T5Config, UMT5Config, UMT5EncoderModel

    torch.manual_seed(5)
    te = UMT5EncoderModel(UMT5Config(vocab_size=101, d_model=24, d_kv=8, d_ff=48, num_layers=2, num_heads=3)).eval()
    tok = HashTokenizer(101, 512)
    with torch.no_grad():
        out = plugin.encode_prompts_wan(tok, te, ["a  red &amp;amp; blue   fox", "two words", ""], max_sequence_length=32)
    assert tuple(out.shape) == (3, 32, 24)
    lens = [len("a red & blue fox".split()) + 1, 3, 1]  # tokens + eos of the CLEANED text (html-unescaped twice, whitespace collapsed)
    for row, n in zip(out, lens):
        assert float(row[n:].abs().max()) == 0.0 and float(row[:n].abs().min(dim=1).values.max()) > 0.0
    wan = plugin.Wan21MI355Model("cpu", dtype=torch.float32)
    wan.tokenizer, wan.text_encoder = tok, te
    pe = wan.get_prompt_embeds("two words")
    assert tuple(pe.text_embeds.shape) == (1, 512, 24) and pe.pooled_embeds is None
    del T5Config


def _clips(seed=41):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection

    torch.manual_seed(seed)
    ccfg = dict(vocab_size=99, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2, max_position_embeddings=16, eos_token_id=1, pad_token_id=0, bos_token_id=2)
    c1 = CLIPTextModel(CLIPTextConfig(hidden_size=32, **ccfg)).eval()
    c2 = CLIPTextModelWithProjection(CLIPTextConfig(hidden_size=48, projection_dim=40, **ccfg)).eval()
    return [HashTokenizer(99, 16), HashTokenizer(99, 16)], [c1, c2]


def test_sd_and_sdxl_prompt_encoding_equal_the_reference_functions():
    """toolkit/train_tools.py:192-323, 379-422 (encode_prompts / encode_prompts_xl incl. long prompts in windows, a disabled first encoder,
    num_images_per_prompt) executed on tiny CLIP encoders — the plug-in's restatements land on the same tensors"""
    gold = load_file(GOLD)
    tk, (c1, c2) = _clips()
    long_prompts = [" ".join(f"w{i}" for i in range(40)), "short one"]
    with torch.no_grad():
        assert torch.equal(plugin.encode_prompts(tk[0], c1, list(PROMPTS)), gold["sd/plain"])
        for tag, kw in (("plain", {}), ("no_te1", {"use_text_encoder_1": False}), ("two_images", {"num_images_per_prompt": 2})):
            e, p = plugin.encode_prompts_xl(tk, [c1, c2], list(PROMPTS), None, **kw)
            assert torch.equal(e, gold[f"sdxl/{tag}/embeds"]) and torch.equal(p, gold[f"sdxl/{tag}/pooled"]), tag
        e, p = plugin.encode_prompts_xl(tk, [c1, c2], long_prompts, None, truncate=False, max_length=64)
        assert torch.equal(e, gold["sdxl/long/embeds"]) and torch.equal(p, gold["sdxl/long/pooled"])
    assert gold["sdxl/plain/embeds"].shape == (3, 16, 32 + 48) and gold["sdxl/plain/pooled"].shape == (3, 40)
    assert gold["sdxl/long/embeds"].shape[1] == 48 and gold["sdxl/two_images/embeds"].shape[0] == 6
    # the plug-in hook (toolkit/stable_diffusion_model.py encode_prompt, sd1 / sdxl branches)
    for xl in (False, True):
        sd = plugin.StableDiffusionMI355Model("cpu", dtype=torch.float32, is_xl=xl)
        sd.tokenizer, sd.text_encoder = (tk, [c1, c2]) if xl else (tk[0], c1)
        pe = sd.get_prompt_embeds(list(PROMPTS))
        if xl:
            assert torch.equal(pe.text_embeds, gold["sdxl/plain/embeds"]) and torch.equal(pe.pooled_embeds, gold["sdxl/plain/pooled"])
        else:
            assert torch.equal(pe.text_embeds, gold["sd/plain"]) and pe.pooled_embeds is None

"""``LattePipeline`` of ``sample/pipeline_latte.py`` (:100-115 constructor, :516-798 ``__call__``) around the MI355X
denoiser: the sampling loop of Latte-1 text-to-video.

What runs where: the transformer call of every step (``latte_amd.LatteT2V``, the hot path) and the VAE decode
(``latte_amd.AutoencoderKL``) run on the HIP engine.  The text encoder / tokenizer are whatever the caller passes (a
``transformers`` T5 in the reference) or are bypassed with ``prompt_embeds`` / ``negative_prompt_embeds``; the scheduler is
any object with the diffusers interface (``latte_amd.schedulers.DDIMScheduler`` is a self-contained stand-in).  With that
scheduler at eta = 0 the whole guided loop (:700-760) runs inside the engine (``latte_t2v_guided_ddim_loop``: text context
computed once, guidance combine + learned-sigma drop + DDIM update fused into one kernel per step); with any other
scheduler object the guidance combine, the learned-sigma drop and the scheduler update are the reference's own few
elementwise lines (:747-758) on device tensors around the engine denoiser.  ``enable_vae_temporal_decoder=True`` decodes through
``latte_amd.AutoencoderKLTemporalDecoder`` in chunks of 14 frames (:779-798), else per frame (:765-777).
"""
import inspect

import torch

from ._lib import LatteError


class VideoPipelineOutput:
    def __init__(self, video):
        self.video = video


class LattePipeline:
    def __init__(self, tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None):
        if transformer is None or scheduler is None:
            raise LatteError("LattePipeline needs at least a transformer (latte_amd.LatteT2V) and a scheduler")
        self.tokenizer, self.text_encoder, self.vae, self.transformer, self.scheduler = tokenizer, text_encoder, vae, transformer, scheduler
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8   # pipeline_latte.py:114
        self._device = getattr(transformer, "_device", torch.device("cuda"))

    def to(self, device):
        self._device = torch.device(device)
        for m in (self.transformer, self.vae, self.text_encoder):
            if m is not None and hasattr(m, "to"):
                m.to(device)
        return self

    # ------------------------------------------------------------------ pipeline_latte.py:117-126
    @staticmethod
    def mask_text_embeddings(emb, mask):
        if emb.shape[0] == 1:
            keep_index = int(mask.sum().item())
            return emb[:, :, :keep_index, :], keep_index
        return emb * mask[:, None, :, None], emb.shape[2]

    # ------------------------------------------------------------------ pipeline_latte.py:359-379
    @staticmethod
    def _text_preprocessing(text, clean_caption=False):
        """What the reference does to every prompt AND negative prompt before the tokenizer (T5 is case sensitive, so
        skipping it changes the token ids): ``text.lower().strip()``, or -- with ``clean_caption`` and both ``bs4`` and
        ``ftfy`` importable -- diffusers' DeepFloyd-IF ``_clean_caption`` twice.  That cleaner (a battery of url / html /
        CJK / punctuation regexes, pipeline_latte.py:384-496) is not restated here: ``clean_caption=True`` warns and takes
        the ``lower().strip()`` path, which is also what the reference itself does whenever one of the two packages is
        missing (:360-368) -- the case in this image."""
        if clean_caption:
            import warnings
            warnings.warn("latte_amd.LattePipeline: clean_caption=True is not available (diffusers' _clean_caption is not "
                          "restated); prompts are lower-cased and stripped, as the reference does without bs4 / ftfy")
        if not isinstance(text, (tuple, list)):
            text = [text]
        return [str(t).lower().strip() for t in text]

    # ------------------------------------------------------------------ pipeline_latte.py:127-270
    def encode_prompt(self, prompt, do_classifier_free_guidance=True, negative_prompt="", num_images_per_prompt=1, device=None,
                      prompt_embeds=None, negative_prompt_embeds=None, clean_caption=False, mask_feature=True):
        embeds_initially_provided = prompt_embeds is not None and negative_prompt_embeds is not None
        device = device or self._device
        if prompt is not None and isinstance(prompt, str):
            prompt = [prompt]
        batch_size = len(prompt) if prompt is not None else prompt_embeds.shape[0]
        max_length = 120
        if prompt_embeds is None:
            if self.tokenizer is None or self.text_encoder is None:
                raise LatteError("pass prompt_embeds / negative_prompt_embeds, or construct the pipeline with a tokenizer and a "
                                 "text encoder (T5 in the reference)")
            prompt = self._text_preprocessing(prompt, clean_caption=clean_caption)                 # :182
            ti = self.tokenizer(prompt, padding="max_length", max_length=max_length, truncation=True, return_attention_mask=True,
                                add_special_tokens=True, return_tensors="pt")
            attention_mask = ti.attention_mask.to(device)
            prompt_embeds_attention_mask = attention_mask
            prompt_embeds = self.text_encoder(ti.input_ids.to(device), attention_mask=attention_mask)[0]
        else:
            prompt_embeds_attention_mask = torch.ones_like(prompt_embeds)
        prompt_embeds = prompt_embeds.to(device=device, dtype=torch.float32)
        bs_embed, seq_len, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(bs_embed * num_images_per_prompt, seq_len, -1)
        prompt_embeds_attention_mask = prompt_embeds_attention_mask.view(bs_embed, -1).repeat(num_images_per_prompt, 1)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            uncond_tokens = self._text_preprocessing([negative_prompt] * batch_size, clean_caption=clean_caption)   # :230-231
            ui = self.tokenizer(uncond_tokens, padding="max_length", max_length=prompt_embeds.shape[1],
                                truncation=True, return_attention_mask=True, add_special_tokens=True, return_tensors="pt")
            negative_prompt_embeds = self.text_encoder(ui.input_ids.to(device), attention_mask=ui.attention_mask.to(device))[0]
        if do_classifier_free_guidance:
            seq_len = negative_prompt_embeds.shape[1]
            negative_prompt_embeds = negative_prompt_embeds.to(device=device, dtype=torch.float32)
            negative_prompt_embeds = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1).view(
                batch_size * num_images_per_prompt, seq_len, -1)
        else:
            negative_prompt_embeds = None
        if mask_feature and not embeds_initially_provided:
            masked, keep = self.mask_text_embeddings(prompt_embeds.unsqueeze(1), prompt_embeds_attention_mask)
            neg = negative_prompt_embeds[:, :keep, :] if negative_prompt_embeds is not None else None
            return masked.squeeze(1), neg
        return prompt_embeds, negative_prompt_embeds

    def _fused_loop(self, steps, do_cfg, eta, callback, latent_channels):
        """(timesteps, alpha_t, alpha_prev) when the loop can run inside the engine (latte_t2v_guided_ddim_loop): guidance on,
        the self-contained DDIM scheduler at eta = 0 without clipping, the engine transformer, no per-step callback.  Any
        other scheduler object / configuration takes the step-by-step loop below."""
        from .schedulers import DDIMScheduler
        from .t2v import LatteT2V
        sch, tr = self.scheduler, self.transformer
        if not (do_cfg and callback is None and eta == 0.0 and type(sch) is DDIMScheduler and not sch.clip_sample
                and isinstance(tr, LatteT2V) and tr.config.out_channels in (latent_channels, 2 * latent_channels)
                and getattr(self, "allow_fused_loop", True)):
            return None
        ts = [int(t) for t in steps]
        ratio = sch.num_train_timesteps // sch.num_inference_steps
        a_t = [float(sch.alphas_cumprod[t]) for t in ts]
        a_p = [float(sch.alphas_cumprod[t - ratio]) if t - ratio >= 0 else float(sch.final_alpha_cumprod) for t in ts]
        return ts, a_t, a_p

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None else device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32).to(device)
        else:
            latents = latents.to(device=device, dtype=torch.float32)
        return latents * self.scheduler.init_noise_sigma                                       # pipeline_latte.py:513

    # ------------------------------------------------------------------ pipeline_latte.py:773-785
    def decode_latents(self, latents):
        if self.vae is None:
            raise LatteError("decode_latents needs a VAE (latte_amd.AutoencoderKL)")
        b, c, f, h, w = latents.shape
        z = (1.0 / self.vae.config.scaling_factor) * latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        video = self.vae.decode(z.contiguous()).sample                                         # all frames in one engine call
        video = video.reshape(b, f, *video.shape[1:]).permute(0, 1, 3, 4, 2)                    # '(b f) c h w -> b f h w c'
        return ((video / 2.0 + 0.5).clamp(0, 1) * 255).to(dtype=torch.uint8).cpu().contiguous()

    def decode_latents_with_temporal_decoder(self, latents):
        """pipeline_latte.py:779-798: chunks of 14 frames through ``vae.decode(chunk, num_frames=len(chunk))``."""
        if not getattr(self.vae, "_TEMPORAL", False):
            raise LatteError("decode_latents_with_temporal_decoder needs a latte_amd.AutoencoderKLTemporalDecoder")
        b, c, f, h, w = latents.shape
        z = (1.0 / self.vae.config.scaling_factor) * latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        video = []
        decode_chunk_size = 14
        for i in range(0, z.shape[0], decode_chunk_size):
            chunk = z[i:i + decode_chunk_size].contiguous()
            video.append(self.vae.decode(chunk, num_frames=chunk.shape[0]).sample)
        video = torch.cat(video)
        video = video.reshape(b, f, *video.shape[1:]).permute(0, 1, 3, 4, 2)
        return ((video / 2.0 + 0.5).clamp(0, 1) * 255).to(dtype=torch.uint8).cpu().contiguous()

    # ------------------------------------------------------------------ pipeline_latte.py:516-771
    @torch.no_grad()
    def __call__(self, prompt=None, negative_prompt="", num_inference_steps=20, timesteps=None, guidance_scale=4.5,
                 num_images_per_prompt=1, video_length=None, height=None, width=None, eta=0.0, generator=None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, output_type="pil", return_dict=True, callback=None,
                 callback_steps=1, clean_caption=True, mask_feature=True, enable_temporal_attentions=True,
                 enable_vae_temporal_decoder=False):
        if enable_vae_temporal_decoder and not getattr(self.vae, "_TEMPORAL", False):
            raise LatteError("enable_vae_temporal_decoder=True needs a latte_amd.AutoencoderKLTemporalDecoder as the pipeline's vae "
                             "(sample_t2x.py:31-32 loads it from subfolder 'vae_temporal_decoder')")
        cfg = self.transformer.config
        height = height or cfg.sample_size * self.vae_scale_factor
        width = width or cfg.sample_size * self.vae_scale_factor
        video_length = video_length or cfg.video_length
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None:
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        device = self._device
        do_cfg = guidance_scale > 1.0
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, do_cfg, negative_prompt=negative_prompt, num_images_per_prompt=num_images_per_prompt, device=device,
            prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, clean_caption=clean_caption,
            mask_feature=mask_feature)
        if do_cfg:
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)             # :647
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        steps = self.scheduler.timesteps
        latent_channels = cfg.in_channels
        latents = self.prepare_latents(batch_size * num_images_per_prompt, latent_channels, video_length, height, width, device,
                                       generator, latents)
        extra = {}
        params = set(inspect.signature(self.scheduler.step).parameters.keys())
        if "eta" in params:
            extra["eta"] = eta
        if "generator" in params:
            extra["generator"] = generator
        added_cond_kwargs = {"resolution": None, "aspect_ratio": None}
        fused = self._fused_loop(steps, do_cfg, eta, callback, latent_channels)
        if fused is not None:
            # the whole guided DDIM chain inside the engine: no torch op between the first and the last step
            self.transformer.set_text(prompt_embeds)
            latents = self.transformer.guided_ddim_loop(latents, fused[0], fused[1], fused[2], guidance_scale,
                                                        enable_temporal_attentions)
            steps = []
        for i, t in enumerate(steps):
            latent_model_input = torch.cat([latents] * 2) if do_cfg else latents
            latent_model_input = self.scheduler.scale_model_input(latent_model_input, t)
            current = t if torch.is_tensor(t) else torch.tensor([t], dtype=torch.int64, device=device)
            current = current.reshape(-1)[:1].to(device).expand(latent_model_input.shape[0])
            noise_pred = self.transformer(latent_model_input, encoder_hidden_states=prompt_embeds, timestep=current,
                                          added_cond_kwargs=added_cond_kwargs,
                                          enable_temporal_attentions=enable_temporal_attentions, return_dict=False)[0]
            if do_cfg:
                uncond, text = noise_pred.chunk(2)
                noise_pred = uncond + guidance_scale * (text - uncond)                           # :748-749
            if cfg.out_channels // 2 == latent_channels:
                noise_pred = noise_pred.chunk(2, dim=1)[0]                                        # learned sigma dropped, :752-753
            latents = self.scheduler.step(noise_pred, t, latents, **extra, return_dict=False)[0]
            if callback is not None and i % callback_steps == 0:
                callback(i // getattr(self.scheduler, "order", 1), t, latents)
        if output_type == "latents":
            video = latents
        elif enable_vae_temporal_decoder:                                                 # pipeline_latte.py:737-740
            video = self.decode_latents_with_temporal_decoder(latents)
        else:
            video = self.decode_latents(latents)
        return VideoPipelineOutput(video=video) if return_dict else (video,)

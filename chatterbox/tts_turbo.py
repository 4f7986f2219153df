"""ChatterboxTurboTTS (GPT-2 T3 backbone + 2-step meanflow S3Gen, reference tts_turbo.py) -- scope row a7 of
SURVEY.md section 8, scheduled after the Multilingual path: the meanflow CFM and the sampler's turbo processor order
are already in the kernels (FlowEngine(meanflow=True), cbx_t3_sample order=1); the GPT-2 decode layer is not."""


class ChatterboxTurboTTS:
    @classmethod
    def from_pretrained(cls, device, nano=False):
        raise NotImplementedError("Turbo/Nano GPT-2 T3 backbone is not built yet (SURVEY.md section 8 row a7)")

    from_local = from_pretrained

"""Routing tuples the Solver uses to reshape batches; the data (membership of each model name) is the
reference's src/model/types.py:1-6 -- it is part of the Solver contract (src/solver.py:443-458)."""
MULTI_SPEECH_SEPERATION_MODELS = ("demucs", "conv-tasnet", "rnn-stft-mask")
MULTI_CHANNEL_SEPERATION_MODELS = ("demucs", "conv-tasnet", "unet")
MONARCH_SPEECH_SEPARTAION_MODELS = ("mel-rnn", "dcunet", "crn", "dnn", "unet", "dccrn", "wav-unet")
STFT_MODELS = ("mel-rnn", "dcunet", "crn", "dnn", "unet", "rnn-stft-mask")
WAV_MODELS = ("dccrn", "demucs", "conv-tasnet", "wav-unet")

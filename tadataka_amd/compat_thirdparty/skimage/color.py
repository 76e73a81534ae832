from tadataka_amd import ops


def rgb2gray(rgb):
    """Luminance 0.2125 R + 0.7154 G + 0.0721 B; 2-D input passes through."""
    return ops.rgb2gray(rgb)

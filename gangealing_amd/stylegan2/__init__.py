from .networks import Generator, ModulatedConv2d, StyledConv, ToRGB, ConvLayer, ResBlock, EqualLinear, EqualConv2d, Blur

__all__ = ['Generator', 'ModulatedConv2d', 'StyledConv', 'ToRGB', 'ConvLayer', 'ResBlock', 'EqualLinear',
           'EqualConv2d', 'Blur']

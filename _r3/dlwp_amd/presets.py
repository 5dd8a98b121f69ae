"""The reference's canonical architectures as build_model() input: (layer_name, args, kwargs) triples written exactly as the
reference scripts write them (examples/train.py:142-221, Azure/train_tf.py:208-268).  Used by bench.py, the smoke test,
the examples and the tests; pure data, no device work."""

CF = {'data_format': 'channels_first'}


def _block(k, filters, ks, dil, act):
    return [('PeriodicPadding2D', ((0, k),), dict(CF)),
            ('ZeroPadding2D', ((k, 0),), dict(CF)),
            ('Conv2D', (filters, ks), dict(CF, dilation_rate=dil, padding='valid', activation=act))]


def unet_layers(cs, widths=(32, 64, 128, 64, 32), cout=None, latitude_dependent=False):
    """Sequential U-Net of Azure/train_tf.py:208-268 / examples/train.py:159-219 (non-recurrent part).
    latitude_dependent: the output layer is DLWP.custom.RowConnected2D instead of Conv2D, the switch of
    examples/train_functional.py:53, 191-196."""
    cout = cs[0] if cout is None else cout
    w1, w2, w3, w4, w5 = widths
    layers = _block(2, w1, 3, 2, 'tanh')
    layers[0][2]['input_shape'] = tuple(cs)
    layers += [('MaxPooling2D', (2,), dict(CF))] + _block(1, w2, 3, 1, 'tanh')
    layers += [('MaxPooling2D', (2,), dict(CF))] + _block(1, w3, 3, 1, 'tanh')
    layers += [('UpSampling2D', (2,), dict(CF))] + _block(1, w4, 3, 1, 'tanh')
    layers += [('UpSampling2D', (2,), dict(CF))] + _block(2, w5, 3, 2, 'tanh')
    if latitude_dependent:
        layers += [('PeriodicPadding2D', ((0, 2),), dict(CF)), ('ZeroPadding2D', ((2, 0),), dict(CF)),
                   ('RowConnected2D', (cout, 5), dict(CF, padding='valid', activation='linear'))]
    else:
        layers += _block(2, cout, 5, 1, 'linear')
    return tuple(layers)


def cnn2_layers(cs, hidden=32):
    """Config 1: 2 x (PeriodicPadding2D + ZeroPadding2D + Conv2D 5x5) -- mirrors examples/train.py:159-169,211-219."""
    layers = _block(2, hidden, 5, 1, 'tanh')
    layers[0][2]['input_shape'] = tuple(cs)
    layers += _block(2, cs[0], 5, 1, 'linear')
    return tuple(layers)


def lstm_unet_layers(cs, lstm_mult=4, widths=(32, 64, 128, 64, 32)):
    """examples/train.py:142-221 with model_is_recurrent=True: ConvLSTM2D front end on the (T, C, H, W) input, Reshape to
    (4*T*C, H, W), the sequential U-Net, Reshape back to (T, C, H, W)."""
    t, c, h, w = cs
    cf5 = {'data_format': 'channels_first'}
    front = [('PeriodicPadding3D', ((0, 0, 2),), dict(cf5, input_shape=tuple(cs))),
             ('ZeroPadding3D', ((0, 2, 0),), dict(cf5)),
             ('ConvLSTM2D', (lstm_mult * c, 3), dict(cf5, dilation_rate=2, padding='valid', activation='tanh',
                                                     return_sequences=True)),
             ('Reshape', ((lstm_mult * t * c, h, w),), None)]
    body = list(unet_layers((lstm_mult * t * c, h, w), widths=widths, cout=t * c))
    body[0] = (body[0][0], body[0][1], {k: v for k, v in body[0][2].items() if k != 'input_shape'})
    return tuple(front + body + [('Reshape', ((t, c, h, w),), None)])

"""keras.regularizers as the reference uses them: `l2(lambda_)` on the ConvLSTM2D kernel (examples/train.py:23,154).  The
trainer adds lambda * sum(w^2) to the loss and 2 * lambda * w to the kernel gradient (Keras' definition)."""


class L1L2(object):
    def __init__(self, l1=0., l2=0.):
        if l1:
            raise NotImplementedError('l1 regularisation is not implemented (l2 is)')
        self.l1, self.l2 = float(l1), float(l2)

    def get_config(self):
        return {'l1': self.l1, 'l2': self.l2}


def l2(l=0.01):
    return L1L2(l2=l)


def l1_l2(l1=0.01, l2=0.01):
    return L1L2(l1=l1, l2=l2)

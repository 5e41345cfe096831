"""Training-loop helper kept for script compatibility (reference: stag/utils.py:1-26)."""
import copy


class EarlyStopping:
    """Stop after `patience` consecutive calls in which no tracked loss improved; keep an
    in-memory copy of the state_dict whenever every loss is at its best."""

    def __init__(self, patience=10):
        self.patience = patience
        self.best_losses = None
        self.best_state = None
        self.counter = 0

    def __call__(self, losses, model):
        if self.best_losses is None:
            self.best_losses, self.counter = list(losses), 0
            return False
        better = [l <= b for l, b in zip(losses, self.best_losses)]
        if any(better):
            if all(better):
                self.best_state = copy.deepcopy(model.state_dict())
            self.best_losses = [min(l, b) for l, b in zip(losses, self.best_losses)]
            self.counter = 0
            return False
        self.counter += 1
        return self.counter == self.patience

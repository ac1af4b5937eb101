"""TensorFlow checkpoint 'bundle' reader (tensor_bundle V2) -- see DESIGN.md."""
import os


def latest_checkpoint(model_dir):
    """tf.train.latest_checkpoint (chiron_eval.py:276): parse the text proto
    `checkpoint` file, return the prefix path or None."""
    path = os.path.join(model_dir, "checkpoint")
    if not os.path.exists(path):
        return None
    for line in open(path):
        line = line.strip()
        if line.startswith("model_checkpoint_path:"):
            name = line.split(":", 1)[1].strip().strip('"')
            return name if os.path.isabs(name) else os.path.join(model_dir, name)
    return None

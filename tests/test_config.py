import pytest

from probnmn.config import Config


def test_defaults_yaml_and_overrides(tmp_path):
    c = Config()
    assert c.PHASE == "joint_training" and c.NMN.MODULE_CHANNELS == 128 and c.OPTIM.BATCH_SIZE == 256
    assert c.NMN.IMAGE_FEATURE_SIZE == [1024, 14, 14] and c.DELTA == 0.99
    y = tmp_path / "module_training.yml"
    y.write_text("PHASE: module_training\nOPTIM:\n  BATCH_SIZE: 128\n  LR_INITIAL: 0.0001\n  LR_PATIENCE: 1000000\n")
    c = Config(str(y), ["OPTIM.BATCH_SIZE", 32, "NMN.IMAGE_FEATURE_SIZE", "[1024, 28, 28]", "BETA", "0.7"])
    assert c.PHASE == "module_training" and c.OPTIM.BATCH_SIZE == 32 and c.OPTIM.LR_INITIAL == 1e-4
    assert c.NMN.IMAGE_FEATURE_SIZE == [1024, 28, 28] and c.BETA == 0.7
    with pytest.raises(AttributeError):
        c.BETA = 0.1
    with pytest.raises(AttributeError):
        c.OPTIM.BATCH_SIZE = 1
    with pytest.raises(KeyError):
        Config(None, ["NO.SUCH_KEY", 1])
    out = tmp_path / "dump.yml"
    c.dump(str(out))
    assert Config(str(out)).OPTIM.BATCH_SIZE == 32


def test_from_config_reads_the_reference_keys(tmp_path):
    from probnmn.vocabulary import Vocabulary

    Vocabulary.clevr().save_to_files(str(tmp_path / "vocab"))
    c = Config(None, ["DATA.VOCABULARY", str(tmp_path / "vocab")])
    from probnmn.models import ProgramGenerator, ProgramPrior, QuestionReconstructor

    assert ProgramGenerator.from_config(c)._max_decoding_steps == 26
    assert QuestionReconstructor.from_config(c)._max_decoding_steps == 45
    assert ProgramPrior.from_config(c)._encoder._module.hidden_size == 256

"""Run by tests/test_dropin_binding.py in its own process (it monkey-patches torch and changes directory):
applies the reference-side binding of INTEGRATION.md section 1 -- the code blocks of that file, verbatim -- to the
reference's own loader (/root/reference/models/__init__.py, options/base_options.py) and reports what the loader finds.
Nothing of the reference is copied or modified: the binding modules are created in memory."""
import json
import os
import re
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def integration_blocks():
    """The python code blocks of INTEGRATION.md section 1, in order."""
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    sec = text.split("## 1.")[1].split("## 2.")[0]
    return re.findall(r"```python\n(.*?)```", sec, flags=re.S)


def main():
    from tests.golden.make_golden import install_shims, ref_options
    install_shims()                                  # torchvision stub, .cuda() -> identity, cwd = the reference
    blocks = integration_blocks()
    model_block = [b for b in blocks if "REGISTRATIONModel" in b][0]
    register_block = [b for b in blocks if "BaseModel.register" in b][0]
    factory_block = [b for b in blocks if "define_G" in b][0]

    import models                                    # the reference's package (models/__init__.py)
    # `models/registration_model.py -- replace the body with:` the re-export
    mod = types.ModuleType("models.registration_model")
    mod.__file__ = "<INTEGRATION.md section 1>"
    exec(compile(model_block, mod.__file__, "exec"), mod.__dict__)
    sys.modules["models.registration_model"] = mod
    models.registration_model = mod
    # `models/__init__.py, after the BaseModel import`
    exec(compile(register_block, "<INTEGRATION.md section 1>", "exec"), models.__dict__)
    ns = {}
    exec(compile(factory_block, "<INTEGRATION.md section 1>", "exec"), ns)   # the factory re-exports import cleanly

    out = {}
    cls = models.find_model_using_name("registration")
    import dfmir_amd.registration_model as hip
    out["found_hip_class"] = cls is hip.REGISTRATIONModel
    out["option_setter_is_hip"] = models.get_option_setter("registration") == hip.REGISTRATIONModel.modify_commandline_options
    opt = ref_options(64, 2, 8)                      # TrainOptions().parse() through the reference's two-pass gather
    out["opt_model"] = opt.model
    out["nce_idt"], out["lambda_NCE"], out["pool_size"] = bool(opt.nce_idt), float(opt.lambda_NCE), int(opt.pool_size)
    # every `opt.<field>` the package reads (not through getattr) must exist on the reference's parsed options
    missing, read = [], set()
    pkg = os.path.join(REPO, "dfmir_amd")
    for fn in sorted(os.listdir(pkg)):
        if not fn.endswith(".py") or fn in ("options.py", "train.py", "test.py"):   # (those BUILD their options, they do not consume the reference's)
            continue
        import ast
        for node in ast.walk(ast.parse(open(os.path.join(pkg, fn)).read())):      # attribute READS in code, not prose
            if not (isinstance(node, ast.Attribute) and isinstance(node.ctx, ast.Load)):
                continue
            v = node.value
            is_opt = (isinstance(v, ast.Name) and v.id == "opt") or (
                isinstance(v, ast.Attribute) and v.attr == "opt" and isinstance(v.value, ast.Name) and v.value.id == "self")
            if is_opt:
                read.add(node.attr)
                if not hasattr(opt, node.attr):
                    missing.append("%s:%s" % (fn, node.attr))
    out["fields_read"] = sorted(read)
    out["missing_fields"] = sorted(set(missing))
    out["factories"] = sorted(k for k in ns if not k.startswith("__"))
    print("DROPIN_RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
